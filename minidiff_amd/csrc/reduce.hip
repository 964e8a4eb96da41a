// reduce.hip — sum / prod / max / min / any / all / argmax / argmin on gfx950,
// including the reduce-to-shape sums of the broadcast-gradient path.
//
// Serves reference minidiff/backend/numpy.py:20-23,43-46,56-57 (np.sum & co.) and
// therefore minidiff/ops/definitions.py:157-183 (unbroadcast_forward: sum over
// prepended / stretched axes — e.g. the bias gradient sum(g, axis=(0,)) of
// SURVEY.md §8 a7, a column reduction) and :224-262 / :192-206.
//
// All HBM-bound: each input element is read exactly once. Three shapes of walk,
// chosen per call from the collapsed (kept | reduced) plan:
//   rows : a block owns one output; lanes stride the reduced extent (16-B loads
//          when it is contiguous), wave shuffles -> LDS -> lane 0. Long rows are
//          split over many blocks into a partial buffer + a finishing pass, so a
//          full reduce of 1e8 elements still fills 256 CUs. Deterministic: no
//          atomics, fixed combine order.
//   cols : a lane owns one output column and walks the reduced rows, neighbouring
//          lanes read neighbouring addresses (coalesced); rows are split over
//          gridDim.y into partials [split][n_out] + a finishing pass.
//   generic : one lane per output, div/mod walk — small or oddly strided inputs.
#include "md_hip.h"
#include "md_narrow.h"

extern "C" int mdhip_alloc(size_t, void **);
extern "C" int mdhip_free(void *);

namespace {

template <class R, class T> __device__ __forceinline__ T md_wave_reduce(T v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = R::combine(v, md_shfl_down(v, d));
  return v;
}
// result valid in thread 0
template <class R, class T> __device__ __forceinline__ T md_block_reduce(T v, T *smem) {
  v = md_wave_reduce<R>(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) smem[w] = v;
  __syncthreads();
  if (w == 0) {
    v = lane < nw ? smem[lane] : R::template identity<T>();
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) v = R::combine(v, md_shfl_down(v, d));
  }
  return v;
}

// ------------------------------------------------------------------- rows ------
// 16-B load with a non-temporal hint (operands larger than the Infinity Cache are read once)
template <bool NT, class V> __device__ __forceinline__ V md_ld_once(const V *p) {
  if constexpr (NT && sizeof(V) == 16) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 t = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(p));
    V v;
    __builtin_memcpy(&v, &t, 16);
    return v;
  } else {
    return *p;
  }
}

// MODE 0: one block per output, dst = the output. MODE 1: `splits` blocks per output leave partials in dst (finished by
// k_finish_rows: 1-byte accumulators). MODE 2: partials + ticket (md_ticket.h): the block that arrives last at its output's
// counter sums the output's partials in index order — the finishing pass without its launch (4.7 us + the launch boundary
// on cfg4's 128 MiB loss sum; same summation order as k_finish_rows, so the results did not change).
template <class R, class Tacc, class Tdst, int MODE, bool NT = false>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_rows(MdRedPlan pl, const void *x, int xdt, int64_t splits, Tdst *dst, Tacc *partial, unsigned *tickets) {
  __shared__ Tacc smem[MD_BLOCK / 64];
  __shared__ unsigned last_flag;
  const int64_t b = blockIdx.x;
  const int64_t o = b / splits, s = b - o * splits;
  int64_t xo, oo;
  md_red_kept_offsets(pl, o, &xo, &oo);
  // the `splits` blocks of one row sweep it together: lane (s, tid) takes items
  // s*blockDim + tid, + splits*blockDim, ... (neighbouring blocks read neighbouring
  // lines at the same time, like the streaming elementwise kernels)
  const int64_t lane0 = s * blockDim.x + threadIdx.x, step = splits * blockDim.x;
  const int64_t n = pl.n_red;
  Tacc acc = R::template identity<Tacc>();
  constexpr int V = 16 / sizeof(Tacc);
  bool vec = false;
  if constexpr (sizeof(Tacc) >= 4) vec = (pl.nr == 1 && pl.rx[0] == 1 && xdt == md_dtype_of<Tacc>::value);
  if (vec) {
    if constexpr (sizeof(Tacc) >= 4) {
      const Tacc *p = (const Tacc *)x + xo;
      // peel to 16-B alignment, then V-wide loads, two in flight per lane
      int64_t head = (V - (int64_t)(((uintptr_t)p / sizeof(Tacc)) % V)) % V;
      if (head > n) head = n;
      if (lane0 < head) acc = R::combine(acc, p[lane0]);
      const int64_t nvec = (n - head) / V;
      Tacc a2[V], a3[V];
#pragma unroll
      for (int j = 0; j < V; ++j) { a2[j] = R::template identity<Tacc>(); a3[j] = a2[j]; }
      const MdVec<Tacc, V> *pv = reinterpret_cast<const MdVec<Tacc, V> *>(p + head);
      int64_t i = lane0;
      // four loads in flight per lane while the row lasts (a 256-KiB row on two blocks: 2 x 256 lanes x 2 loads left the CU with
      // 16 KiB in flight — the bn-style sum over (0, 2, 3) of 32 x 16 x 256 x 256 ran at 3.7 TB/s), then two, then one
      for (; i + 3 * step < nvec; i += 4 * step) {
        MdVec<Tacc, V> t = md_ld_once<NT>(pv + i);
        MdVec<Tacc, V> u = md_ld_once<NT>(pv + i + step);
        MdVec<Tacc, V> t2 = md_ld_once<NT>(pv + i + 2 * step);
        MdVec<Tacc, V> u2 = md_ld_once<NT>(pv + i + 3 * step);
#pragma unroll
        for (int j = 0; j < V; ++j) { a2[j] = R::combine(a2[j], t.v[j]); a3[j] = R::combine(a3[j], u.v[j]); }
#pragma unroll
        for (int j = 0; j < V; ++j) { a2[j] = R::combine(a2[j], t2.v[j]); a3[j] = R::combine(a3[j], u2.v[j]); }
      }
      for (; i + step < nvec; i += 2 * step) {
        MdVec<Tacc, V> t = md_ld_once<NT>(pv + i);
        MdVec<Tacc, V> u = md_ld_once<NT>(pv + i + step);
#pragma unroll
        for (int j = 0; j < V; ++j) { a2[j] = R::combine(a2[j], t.v[j]); a3[j] = R::combine(a3[j], u.v[j]); }
      }
      if (i < nvec) {
        MdVec<Tacc, V> t = pv[i];
#pragma unroll
        for (int j = 0; j < V; ++j) a2[j] = R::combine(a2[j], t.v[j]);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) acc = R::combine(acc, R::combine(a2[j], a3[j]));
      const int64_t t0 = head + nvec * V;
      if (t0 + lane0 < n) acc = R::combine(acc, p[t0 + lane0]);
    }
  } else if (pl.nr == 1) {
    const int64_t rs = pl.rx[0];
    for (int64_t r = lane0; r < n; r += step) acc = R::combine(acc, md_load<Tacc>(x, xdt, xo + r * rs));
  } else {
    for (int64_t r = lane0; r < n; r += step) acc = R::combine(acc, md_load<Tacc>(x, xdt, xo + md_red_offset(pl, r)));
  }
  acc = md_block_reduce<R>(acc, smem);
  if constexpr (MODE == 0) {
    if (threadIdx.x == 0) dst[oo] = md_cast<Tdst>(acc);
  } else if constexpr (MODE == 1) {
    if (threadIdx.x == 0) dst[b] = md_cast<Tdst>(acc);
  } else {
    if (threadIdx.x == 0) md_st_sc1(partial + b, acc);
    const bool last = splits >= 64 ? md_ticket_last2(tickets + o * MD_TICKET2_WORDS, (unsigned)s, (unsigned)splits, &last_flag)
                                   : md_ticket_last(tickets + o * MD_TICKET_PAD, (unsigned)splits, &last_flag);
    if (!last) return;
    Tacc a = md_fold_partials<R>(partial + o * splits, (unsigned)splits);
    a = md_block_reduce<R>(a, smem);
    if (threadIdx.x == 0) dst[oo] = md_cast<Tdst>(a);
  }
}

// SHORT contiguous rows (softmax / layer-norm style sums over a last axis of <= 64 * V * NV elements): a WAVE per output, four
// outputs per block, all of a lane's 16-B loads issued before the first combine. The block-per-output kernel above gives such a
// row 256 threads for at most a few loads each and one block launch per 4 KiB (64 x 512 x 1024 summed over the last axis: 3.5 TB/s).
template <class R, class Tacc, class Tdst, int NV>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_rows_wave(MdRedPlan pl, const Tacc *__restrict__ x, Tdst *__restrict__ dst) {
  constexpr int V = 16 / sizeof(Tacc);
  typedef MdVec<Tacc, V> Vec;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t o = (int64_t)blockIdx.x * 4 + w;
  if (o >= pl.n_out) return;
  int64_t xo, oo;
  md_red_kept_offsets(pl, o, &xo, &oo);
  const Tacc *p = x + xo;
  const Vec *pv = reinterpret_cast<const Vec *>(p);
  const int64_t n = pl.n_red, nvec = n / V;
  Vec t[NV];
#pragma unroll
  for (int g = 0; g < NV; ++g) {
    const int64_t i = lane + 64 * g;
    if (i < nvec) t[g] = pv[i];
  }
  Tacc acc = R::template identity<Tacc>();
#pragma unroll
  for (int g = 0; g < NV; ++g) {
    if (lane + 64 * g < nvec) {
#pragma unroll
      for (int j = 0; j < V; ++j) acc = R::combine(acc, t[g].v[j]);
    }
  }
  const int64_t t0 = nvec * V;
  if (t0 + lane < n) acc = R::combine(acc, p[t0 + lane]);
  acc = md_wave_reduce<R>(acc);
  if (lane == 0) dst[oo] = md_cast<Tdst>(acc);
}

// .. and the same wave per output for rows the kernel above does not take: 1- and 2-byte element types (any / all of a bool matrix
// over its last axis, sums of int8), rows that do not start on 16-B boundaries, operands of another dtype than the accumulator —
// lanes stride the row element by element through md_load (the dtype switch is wave-uniform). The block-per-output kernel gave a
// 268-byte row 256 threads and its own block: 10^6 such rows of bool took 1.28 ms (210 GB/s).
template <class R, class Tacc, class Tdst>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_rows_wave_any(MdRedPlan pl, const void *x, int xdt, Tdst *__restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t o = wave; o < pl.n_out; o += n_waves) {
    int64_t xo, oo;
    md_red_kept_offsets(pl, o, &xo, &oo);
    Tacc acc = R::template identity<Tacc>();
    for (int64_t r = lane; r < pl.n_red; r += 64) acc = R::combine(acc, md_load<Tacc>(x, xdt, xo + r));
    acc = md_wave_reduce<R>(acc);
    if (lane == 0) dst[oo] = md_cast<Tdst>(acc);
  }
}

// Full reduction of a contiguous array of the accumulator's own type (the loss `sum` of the BASELINE graphs): the rows
// kernel above without its plan — a lane strides the 16-B vectors of the whole grid, two in flight — and the ticket finish
// (md_ticket.h, two-level: ~1000 blocks arrive). 24.0 -> 22.5 us on cfg4's 128 MiB against the general kernel: the plan's
// 400-byte argument block and per-thread offset arithmetic cost 1.5 us of a 20-us stream (profiles/r3_reduce_lab.txt).
// S: the array's storage type — the accumulator's own, or a storage-only dtype (int8 .. float16: 16 / sizeof(S) elements per 16-B
// load, each converted to the accumulator type: sum(int8) reads ONE byte per element)
template <class R, class Tacc, class Tdst, bool NT, class S = Tacc>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_all(const S *__restrict__ x, int64_t n, Tacc *partial, unsigned *tickets, Tdst *out) {
  constexpr int V = 16 / sizeof(S);
  typedef MdVec<S, V> Vec;
  __shared__ Tacc smem[MD_BLOCK / 64];
  __shared__ unsigned last_flag;
  const int64_t gs = (int64_t)gridDim.x * MD_BLOCK, gid = (int64_t)blockIdx.x * MD_BLOCK + threadIdx.x;
  const int64_t nvec = n / V;  // (x is 16-B aligned: checked on the host)
  const Vec *pv = reinterpret_cast<const Vec *>(x);
  Tacc a2[V], a3[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a2[j] = R::template identity<Tacc>(); a3[j] = a2[j]; }
  int64_t i = gid;
  for (; i + gs < nvec; i += 2 * gs) {
    const Vec t = md_ld_once<NT>(pv + i), u = md_ld_once<NT>(pv + i + gs);
#pragma unroll
    for (int j = 0; j < V; ++j) { a2[j] = R::combine(a2[j], md_cast<Tacc>(t.v[j])); a3[j] = R::combine(a3[j], md_cast<Tacc>(u.v[j])); }
  }
  if (i < nvec) {
    const Vec t = pv[i];
#pragma unroll
    for (int j = 0; j < V; ++j) a2[j] = R::combine(a2[j], md_cast<Tacc>(t.v[j]));
  }
  Tacc acc = R::template identity<Tacc>();
#pragma unroll
  for (int j = 0; j < V; ++j) acc = R::combine(acc, R::combine(a2[j], a3[j]));
  if (nvec * V + gid < n) acc = R::combine(acc, md_cast<Tacc>(x[nvec * V + gid]));  // the up-to-(V-1) elements behind the last whole vector
  acc = md_block_reduce<R>(acc, smem);
  if (threadIdx.x == 0) md_st_sc1(partial + blockIdx.x, acc);
  const bool last = gridDim.x >= 64 ? md_ticket_last2(tickets, blockIdx.x, gridDim.x, &last_flag) : md_ticket_last(tickets, gridDim.x, &last_flag);
  if (!last) return;
  Tacc a = md_fold_partials<R>(partial, gridDim.x);
  a = md_block_reduce<R>(a, smem);
  if (threadIdx.x == 0) out[0] = md_cast<Tdst>(a);
}

template <class R, class Tacc, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_finish_rows(MdRedPlan pl, const Tacc *partial, int64_t splits, To *out) {
  __shared__ Tacc smem[MD_BLOCK / 64];
  const int64_t o = blockIdx.x;
  Tacc acc = R::template identity<Tacc>();
  for (int64_t s = threadIdx.x; s < splits; s += blockDim.x) acc = R::combine(acc, partial[o * splits + s]);
  acc = md_block_reduce<R>(acc, smem);
  if (threadIdx.x == 0) {
    int64_t xo, oo;
    md_red_kept_offsets(pl, o, &xo, &oo);
    out[oo] = md_cast<To>(acc);
  }
}

// ------------------------------------------------------------------- cols ------
template <class R, class Tacc, class Tdst, bool FINAL>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_cols(MdRedPlan pl, const void *x, int xdt, int64_t chunk, Tdst *dst) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= pl.n_out) return;
  const int64_t s = blockIdx.y;
  int64_t xo, oo;
  md_red_kept_offsets(pl, o, &xo, &oo);
  const int64_t r0 = s * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > pl.n_red) r1 = pl.n_red;
  Tacc a0 = R::template identity<Tacc>(), a1 = a0, a2 = a0, a3 = a0;
  if (pl.nr == 1) {
    const int64_t rs = pl.rx[0];
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {  // four loads in flight per lane
      Tacc t0 = md_load<Tacc>(x, xdt, xo + (r + 0) * rs);
      Tacc t1 = md_load<Tacc>(x, xdt, xo + (r + 1) * rs);
      Tacc t2 = md_load<Tacc>(x, xdt, xo + (r + 2) * rs);
      Tacc t3 = md_load<Tacc>(x, xdt, xo + (r + 3) * rs);
      a0 = R::combine(a0, t0); a1 = R::combine(a1, t1); a2 = R::combine(a2, t2); a3 = R::combine(a3, t3);
    }
    for (; r < r1; ++r) a0 = R::combine(a0, md_load<Tacc>(x, xdt, xo + r * rs));
  } else {
    for (int64_t r = r0; r < r1; ++r) a0 = R::combine(a0, md_load<Tacc>(x, xdt, xo + md_red_offset(pl, r)));
  }
  Tacc acc = R::combine(R::combine(a0, a1), R::combine(a2, a3));
  if constexpr (FINAL) dst[oo] = md_cast<Tdst>(acc);
  else dst[s * pl.n_out + o] = acc;
}

template <class R, class Tacc, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_finish_cols(MdRedPlan pl, const Tacc *partial, int64_t splits, To *out) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= pl.n_out) return;
  Tacc acc = R::template identity<Tacc>();
  for (int64_t s = 0; s < splits; ++s) acc = R::combine(acc, partial[s * pl.n_out + o]);
  int64_t xo, oo;
  md_red_kept_offsets(pl, o, &xo, &oo);
  out[oo] = md_cast<To>(acc);
}

// --------------------------------------------------------- cols, vectorised ------
// The reduce-to-shape of the broadcast-gradient path at scale (bias gradient:
// sum(g[8192,4096], axis=0)). Block = 64 column groups x 4 row lanes; a lane owns
// V adjacent columns (one 16-B load per row), a wave reads 1 KiB of one row per
// instruction, the four waves of a block walk four different rows, four rows in
// flight per lane (64 B/lane). Row lanes are combined through LDS, row chunks
// through a partial buffer [split][n_out] reduced by the same kernel.
template <class R, class Tacc, class Tdst, bool FINAL>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_cols_vec(const Tacc *__restrict__ x, int64_t n_out, int64_t n_red, int64_t rs,
                                                             int64_t chunk, Tdst *__restrict__ dst) {
  constexpr int V = 16 / sizeof(Tacc);
  __shared__ Tacc smem[3][64][V];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t col = ((int64_t)blockIdx.x * 64 + cx) * V;
  const int64_t s = blockIdx.y;
  const int64_t r0 = s * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > n_red) r1 = n_red;
  Tacc acc[4][V];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[u][j] = R::template identity<Tacc>();
  if (col < n_out) {
    const Tacc *p = x + col;
    int64_t r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      MdVec<Tacc, V> t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const MdVec<Tacc, V> *>(p + (r + 4 * u) * rs);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[u][j] = R::combine(acc[u][j], t[u].v[j]);
    }
    for (; r < r1; r += 4) {
      MdVec<Tacc, V> t = *reinterpret_cast<const MdVec<Tacc, V> *>(p + r * rs);
#pragma unroll
      for (int j = 0; j < V; ++j) acc[0][j] = R::combine(acc[0][j], t.v[j]);
    }
  }
  Tacc tot[V];
#pragma unroll
  for (int j = 0; j < V; ++j) tot[j] = R::combine(R::combine(acc[0][j], acc[1][j]), R::combine(acc[2][j], acc[3][j]));
  if (ry > 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) smem[ry - 1][cx][j] = tot[j];
  }
  __syncthreads();
  if (ry == 0 && col < n_out) {
    MdVec<Tdst, V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      Tacc v = R::combine(R::combine(tot[j], smem[0][cx][j]), R::combine(smem[1][cx][j], smem[2][cx][j]));
      o.v[j] = md_cast<Tdst>(v);
    }
    Tdst *d = FINAL ? dst + col : dst + s * n_out + col;
    *reinterpret_cast<MdVec<Tdst, V> *>(d) = o;
  }
}

// Reduce-to-shape at scale (the bias gradient sum(g[8192, 4096], axis=0) of SURVEY §8 a7), ONE launch.
// Grid = NS column strips x NB row bands, a band's blocks are neighbours. A strip is 64 lanes x one 16-B vector (1 KiB
// of a row); the four waves of a block and the NB bands take the rows INTERLEAVED (row = band + NB * (wave + 4 i)), so
// at any moment the whole grid reads one moving window of the matrix, like the streaming kernels — with a contiguous
// band per block the same kernel ran at 5.3 TB/s, interleaved at 6.9 (profiles/r3_reduce_lab.txt). Two batches of RB
// rows are in flight per lane (software pipeline: one wave per SIMD, nothing else covers the issue gap). Waves are
// combined through LDS in wave order, bands through write-through partial rows and a ticket per strip (md_ticket.h):
// the block that arrives last at its strip's counter adds the strip's NB partial rows in band order. No atomics on
// data, fixed order: bit-identical from run to run.
template <class R, class Tacc, int RB, bool NT>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_cols_strips(const Tacc *__restrict__ x, int64_t n_out, int64_t n_red, int64_t rs, int NS,
                                                                int NB, Tacc *partial, unsigned *tickets, Tacc *__restrict__ out,
                                                                int64_t x_bs, int64_t o_bs) {
  constexpr int V = 16 / sizeof(Tacc);
  typedef MdVec<Tacc, V> Vec;
  __shared__ Vec sm[3][64];
  __shared__ unsigned last_flag;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int s = blockIdx.x % NS, b = blockIdx.x / NS;
  // blockIdx.y: one of several independent (n_red x n_out) problems x_bs / o_bs elements apart (a MIDDLE axis reduced: sum over the
  // sequence axis of (batch, seq, hidden)); each has its own partial rows and its own strip tickets
  x += (int64_t)blockIdx.y * x_bs;
  out += (int64_t)blockIdx.y * o_bs;
  if (NB > 1) partial += (int64_t)blockIdx.y * NB * n_out;
  tickets += (int64_t)blockIdx.y * NS * MD_TICKET_PAD;
  const int64_t col_raw = ((int64_t)s * 64 + lane) * V;
  const bool col_ok = col_raw < n_out;
  const int64_t col = col_ok ? col_raw : n_out - V;  // lanes past a ragged edge load a valid vector and store nothing
  const int64_t first = b + (int64_t)NB * w, step = (int64_t)NB * 4;
  const int64_t nrw = first < n_red ? (n_red - first + step - 1) / step : 0;  // rows of this wave: first + step * i
  const int64_t nb = nrw / RB;
  Tacc acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = R::template identity<Tacc>();
  const Tacc *p = x + col + first * rs;
  const int64_t rstep = step * rs;
  Vec t[2][RB];
  auto load = [&](int buf, int64_t bt) {
    const int64_t i0 = (bt < nb ? bt : nb - 1) * RB;  // a prefetch past the end re-reads the last batch (discarded): no branch
#pragma unroll
    for (int u = 0; u < RB; ++u) t[buf][u] = md_ld_once<NT>(reinterpret_cast<const Vec *>(p + (i0 + u) * rstep));
  };
  auto add = [&](int buf) {
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = R::combine(acc[j], t[buf][u].v[j]);
  };
  if (nb > 0) {
    load(0, 0);
    int64_t bt = 0;
    for (; bt + 1 < nb; bt += 2) {
      // (scheduling barriers pin "issue the next batch, THEN consume the landed one"; left alone the compiler sinks the
      // prefetch behind the adds and the queue drains every trip)
      load(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
      add(0);
      __builtin_amdgcn_sched_barrier(0);
      load(0, bt + 2);
      __builtin_amdgcn_sched_barrier(0);
      add(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bt < nb) add(0);  // odd batch count: the last batch sits in buffer 0
  }
  for (int64_t i = nb * RB; i < nrw; ++i) {
    const Vec tt = *reinterpret_cast<const Vec *>(p + i * rstep);
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = R::combine(acc[j], tt.v[j]);
  }
  // waves 1..3 -> wave 0, in wave order
  if (w > 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) sm[w - 1][lane].v[j] = acc[j];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = R::combine(acc[j], sm[k][lane].v[j]);
  }
  if (NB == 1) {
    if (w == 0 && col_ok) {
      Vec o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.v[j] = acc[j];
      *reinterpret_cast<Vec *>(out + col) = o;
    }
    return;
  }
  const __amdgpu_buffer_rsrc_t pr = md_rsrc(partial, (unsigned)((int64_t)NB * n_out * (int64_t)sizeof(Tacc)));
  if (w == 0 && col_ok) {
    Vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = acc[j];
    md_st16_sc1(pr, (unsigned)(((int64_t)b * n_out + col) * (int64_t)sizeof(Tacc)), o);
  }
  if (!md_ticket_last(tickets + s * MD_TICKET_PAD, (unsigned)NB, &last_flag)) return;
  // the strip's NB (<= 64) partial rows: wave w takes rows w, w + 4, ..; every load ahead of the first add
  Vec pt[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = w + 4 * k;
    if (r < NB) pt[k] = md_ld16_sc1<Vec>(pr, (unsigned)(((int64_t)r * n_out + col) * (int64_t)sizeof(Tacc)));
  }
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = R::template identity<Tacc>();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (w + 4 * k < NB) {
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = R::combine(acc[j], pt[k].v[j]);
    }
  }
  if (w > 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) sm[w - 1][lane].v[j] = acc[j];
  }
  __syncthreads();
  if (w == 0 && col_ok) {
    Vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      Tacc v = acc[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) v = R::combine(v, sm[k][lane].v[j]);
      o.v[j] = v;
    }
    *reinterpret_cast<Vec *>(out + col) = o;
  }
}

// ---------------------------------------------------------------- generic ------
template <class R, class Tacc, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_reduce_generic(MdRedPlan pl, const void *x, int xdt, To *out) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < pl.n_out; o += gs) {
    int64_t xo, oo;
    md_red_kept_offsets(pl, o, &xo, &oo);
    Tacc acc = R::template identity<Tacc>();
    for (int64_t r = 0; r < pl.n_red; ++r) acc = R::combine(acc, md_load<Tacc>(x, xdt, xo + md_red_offset(pl, r)));
    out[oo] = md_cast<To>(acc);
  }
}

// ------------------------------------------------------------ arg-reductions ----
template <bool IsMax, class T>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_thread(MdRedPlan pl, const void *x, int xdt, int64_t *out) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < pl.n_out; o += gs) {
    int64_t xo, oo;
    md_red_kept_offsets(pl, o, &xo, &oo);
    md_argpair<T> acc = RArg<IsMax>::template identity<T>();
    for (int64_t r = 0; r < pl.n_red; ++r)
      acc = RArg<IsMax>::combine(acc, md_argpair<T>{md_load<T>(x, xdt, xo + md_red_offset(pl, r)), r});
    out[oo] = acc.i;
  }
}
template <bool IsMax, class T>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_block(MdRedPlan pl, const void *x, int xdt, int64_t *out) {
  __shared__ T sv[MD_BLOCK / 64];
  __shared__ int64_t si[MD_BLOCK / 64];
  const int64_t o = blockIdx.x;
  int64_t xo, oo;
  md_red_kept_offsets(pl, o, &xo, &oo);
  md_argpair<T> acc = RArg<IsMax>::template identity<T>();
  for (int64_t r = threadIdx.x; r < pl.n_red; r += blockDim.x)
    acc = RArg<IsMax>::combine(acc, md_argpair<T>{md_load<T>(x, xdt, xo + md_red_offset(pl, r)), r});
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    md_argpair<T> other{md_shfl_down(acc.v, d), md_shfl_down(acc.i, d)};
    acc = RArg<IsMax>::combine(acc, other);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { sv[w] = acc.v; si[w] = acc.i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < nw; ++k) acc = RArg<IsMax>::combine(acc, md_argpair<T>{sv[k], si[k]});
    out[oo] = acc.i;
  }
}

// Contiguous reduced axis, rows of a few dozen to a thousand elements (max / min backward of a matrix over its last axis):
// a WAVE per row — the lanes stride the row (coalesced), meet their indices in increasing order (so "strictly better" keeps the
// first of equal values) and are merged by RArg::combine. One THREAD per row walked 268 elements 1 KiB apart from its neighbour's:
// 277 GB/s, 3.9 ms for 10^6 rows of 268 where max takes 0.27.
// (TYPED: x holds T itself; otherwise a storage-only dtype read through md_load — argmax of an int8 matrix)
template <bool IsMax, class T, bool TYPED>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_rows_wave(MdRedPlan pl, const void *__restrict__ x, int xdt, int64_t *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t o = wave; o < pl.n_out; o += n_waves) {
    int64_t xo, oo;
    md_red_kept_offsets(pl, o, &xo, &oo);
    md_argpair<T> acc = RArg<IsMax>::template identity<T>();
    for (int64_t r = lane; r < pl.n_red; r += 64) {
      T v;
      if constexpr (TYPED) v = ((const T *)x)[xo + r];
      else v = md_load<T>(x, xdt, xo + r);
      if (acc.i == INT64_MAX || RArg<IsMax>::better(v, acc.v)) { acc.v = v; acc.i = r; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      md_argpair<T> other{md_shfl_down(acc.v, d), md_shfl_down(acc.i, d)};
      acc = RArg<IsMax>::combine(acc, other);
    }
    if (lane == 0) out[oo] = acc.i;
  }
}

// Contiguous reduced axis (argmax over the last axis): one block per output, 16-B loads, two in
// flight per lane. A lane meets its indices in increasing order, so "strictly better" keeps the
// first of equal values (and the first NaN), as np.argmax does; lanes are merged with RArg::combine.
template <bool IsMax, class T, bool FINAL>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_rows_vec(MdRedPlan pl, const T *__restrict__ x, int64_t splits, int64_t chunk,
                                                          T *__restrict__ pval, int64_t *__restrict__ pidx) {
  constexpr int V = 16 / sizeof(T);
  __shared__ T sv[MD_BLOCK / 64];
  __shared__ int64_t si[MD_BLOCK / 64];
  const int64_t b = blockIdx.x;
  const int64_t o = b / splits, s = b - o * splits;
  int64_t xo, oo;
  md_red_kept_offsets(pl, o, &xo, &oo);
  // this block's slice [lo, hi) of the row (long rows are cut into `splits` slices)
  const int64_t lo = s * chunk;
  int64_t hi = lo + chunk;
  if (hi > pl.n_red) hi = pl.n_red;
  const T *p = x + xo + lo;
  const int64_t n = hi - lo;
  md_argpair<T> acc = RArg<IsMax>::template identity<T>();
  // (written as selects: the branchy form `if (first || better) { v = ..; i = ..; }` was observed to
  // lose the update on gfx950 with this compiler)
  auto take = [&](T v, int64_t r) {
    const bool up = (acc.i == INT64_MAX) | RArg<IsMax>::better(v, acc.v);
    acc.v = up ? v : acc.v;
    acc.i = up ? r : acc.i;
  };
  int64_t head = (V - (int64_t)(((uintptr_t)p / sizeof(T)) % V)) % V;
  if (head > n) head = n;
  if ((int64_t)threadIdx.x < head) take(p[threadIdx.x], lo + threadIdx.x);
  const int64_t nvec = (n - head) / V;
  const MdVec<T, V> *pv = reinterpret_cast<const MdVec<T, V> *>(p + head);
  int64_t i = threadIdx.x;
  for (; i + blockDim.x < nvec; i += 2 * blockDim.x) {
    const MdVec<T, V> t = pv[i], u = pv[i + blockDim.x];
#pragma unroll
    for (int j = 0; j < V; ++j) take(t.v[j], lo + head + i * V + j);
#pragma unroll
    for (int j = 0; j < V; ++j) take(u.v[j], lo + head + (i + blockDim.x) * V + j);
  }
  if (i < nvec) {
    const MdVec<T, V> t = pv[i];
#pragma unroll
    for (int j = 0; j < V; ++j) take(t.v[j], lo + head + i * V + j);
  }
  const int64_t t0 = head + nvec * V;
  if (t0 + threadIdx.x < n) take(p[t0 + threadIdx.x], lo + t0 + threadIdx.x);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    md_argpair<T> other{md_shfl_down(acc.v, d), md_shfl_down(acc.i, d)};
    acc = RArg<IsMax>::combine(acc, other);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { sv[w] = acc.v; si[w] = acc.i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < nw; ++k) acc = RArg<IsMax>::combine(acc, md_argpair<T>{sv[k], si[k]});
    if constexpr (FINAL) pidx[oo] = acc.i;
    else { pval[b] = acc.v; pidx[b] = acc.i; }
  }
}
template <bool IsMax, class T>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_rows_finish(MdRedPlan pl, const T *pval, const int64_t *pidx, int64_t splits, int64_t *out) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= pl.n_out) return;
  md_argpair<T> a = RArg<IsMax>::template identity<T>();
  for (int64_t s = 0; s < splits; ++s) a = RArg<IsMax>::combine(a, md_argpair<T>{pval[o * splits + s], pidx[o * splits + s]});
  int64_t xo, oo;
  md_red_kept_offsets(pl, o, &xo, &oo);
  out[oo] = a.i;
}

// Contiguous KEPT axis (argmax over axis 0 of a row-major matrix): a lane owns V adjacent
// columns and walks the rows (coalesced 16-B loads, four rows in flight); the four waves of a
// block take rows r, r+1, r+2, r+3 (mod 4) and are merged through LDS; row chunks (gridDim.y)
// leave (value, index) partials merged by k_arg_cols_finish.
template <bool IsMax, class T, bool FINAL>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_cols_vec(const T *__restrict__ x, int64_t n_out, int64_t n_red, int64_t rs,
                                                          int64_t chunk, T *__restrict__ pval, int64_t *__restrict__ pidx) {
  constexpr int V = 16 / sizeof(T);
  __shared__ T sv[3][64][V];
  __shared__ int64_t si[3][64][V];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t col = ((int64_t)blockIdx.x * 64 + cx) * V;
  const int64_t s = blockIdx.y, r0 = s * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > n_red) r1 = n_red;
  T bv[V];
  int64_t bi[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { bv[j] = T(); bi[j] = INT64_MAX; }
  if (col < n_out) {
    const T *p = x + col;
    int64_t r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      MdVec<T, V> t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const MdVec<T, V> *>(p + (r + 4 * u) * rs);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const bool up = (bi[j] == INT64_MAX) | RArg<IsMax>::better(t[u].v[j], bv[j]);
          bv[j] = up ? t[u].v[j] : bv[j];
          bi[j] = up ? r + 4 * u : bi[j];
        }
    }
    for (; r < r1; r += 4) {
      const MdVec<T, V> t = *reinterpret_cast<const MdVec<T, V> *>(p + r * rs);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const bool up = (bi[j] == INT64_MAX) | RArg<IsMax>::better(t.v[j], bv[j]);
        bv[j] = up ? t.v[j] : bv[j];
        bi[j] = up ? r : bi[j];
      }
    }
  }
  if (ry > 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) { sv[ry - 1][cx][j] = bv[j]; si[ry - 1][cx][j] = bi[j]; }
  }
  __syncthreads();
  if (ry == 0 && col < n_out) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      md_argpair<T> a{bv[j], bi[j]};
#pragma unroll
      for (int k = 0; k < 3; ++k) a = RArg<IsMax>::combine(a, md_argpair<T>{sv[k][cx][j], si[k][cx][j]});
      if constexpr (FINAL) pidx[col + j] = a.i;
      else { pval[s * n_out + col + j] = a.v; pidx[s * n_out + col + j] = a.i; }
    }
  }
}
template <bool IsMax, class T>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_cols_finish(const T *pval, const int64_t *pidx, int64_t n_out, int64_t splits, int64_t *out) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_out) return;
  md_argpair<T> a = RArg<IsMax>::template identity<T>();
  for (int64_t s = 0; s < splits; ++s) a = RArg<IsMax>::combine(a, md_argpair<T>{pval[s * n_out + o], pidx[s * n_out + o]});
  out[o] = a.i;
}

// The same walk as k_reduce_cols_strips for (value, row) pairs: NS strips of 64 lanes x 16 B of columns, NB bands of rows, rows
// interleaved across bands and the block's four waves (a lane meets its rows in increasing order, so "strictly better" keeps the
// first of equal values and the first NaN, as np.argmax does), batches of RB rows double-buffered; band partials (value + row)
// published write-through, merged in band order by the block that arrives last at the strip's ticket. One launch (the chunked
// kernel above + its finish pass ran at 1.5 TB/s on 8192 x 4096: 128 rows per block, four loads in flight per wave).
// Rows are carried as 32-bit (16-byte T) or 64-bit (8-byte T) integers so that a lane's V rows fill one 16-B vector; n_red < 2^31.
template <bool IsMax, class T, int RB>
__global__ void __launch_bounds__(MD_BLOCK) k_arg_cols_strips(const T *__restrict__ x, int64_t n_out, int64_t n_red, int64_t rs, int NS, int NB,
                                                             T *pval, void *pidx_, unsigned *tickets, int64_t *__restrict__ out) {
  constexpr int V = 16 / sizeof(T);
  typedef MdVec<T, V> Vec;
  typedef typename md_cond<V == 4, int32_t, int64_t>::type I;
  typedef MdVec<I, V> IVec;
  static_assert(sizeof(IVec) == 16, "a lane's rows travel as one 16-B vector");
  __shared__ Vec sv[3][64];
  __shared__ IVec si[3][64];
  __shared__ unsigned last_flag;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int s = blockIdx.x % NS, b = blockIdx.x / NS;
  const int64_t col_raw = ((int64_t)s * 64 + lane) * V;
  const bool col_ok = col_raw < n_out;
  const int64_t col = col_ok ? col_raw : n_out - V;
  const int64_t first = b + (int64_t)NB * w, step = (int64_t)NB * 4;
  const int64_t nrw = first < n_red ? (n_red - first + step - 1) / step : 0;
  const int64_t nb = nrw / RB;
  T bv[V];
  I bi[V];
  // bv starts at the identity (lowest / highest value), bi at -1 = "none": an element updates iff it is strictly better — no
  // "first element" test per element. A lane whose rows all EQUAL the identity (a column of -inf) never updates; it is given its
  // first row afterwards, which is what the element-by-element rule would have kept.
#pragma unroll
  for (int j = 0; j < V; ++j) { bv[j] = RArg<IsMax>::template identity<T>().v; bi[j] = (I)-1; }
  // (selects, not branches: see k_arg_rows_vec)
  auto take = [&](const Vec &t, I r) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const bool up = RArg<IsMax>::better(t.v[j], bv[j]);
      bv[j] = up ? t.v[j] : bv[j];
      bi[j] = up ? r : bi[j];
    }
  };
  const T *p = x + col + first * rs;
  const int64_t rstep = step * rs;
  Vec t[2][RB];
  auto load = [&](int buf, int64_t bt) {
    const int64_t i0 = (bt < nb ? bt : nb - 1) * RB;
#pragma unroll
    for (int u = 0; u < RB; ++u) t[buf][u] = *reinterpret_cast<const Vec *>(p + (i0 + u) * rstep);
  };
  auto eat = [&](int buf, int64_t bt) {
#pragma unroll
    for (int u = 0; u < RB; ++u) take(t[buf][u], (I)(first + step * (bt * RB + u)));
  };
  if (nb > 0) {
    load(0, 0);
    int64_t bt = 0;
    for (; bt + 1 < nb; bt += 2) {
      load(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
      eat(0, bt);
      __builtin_amdgcn_sched_barrier(0);
      load(0, bt + 2);
      __builtin_amdgcn_sched_barrier(0);
      eat(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bt < nb) eat(0, bt);
  }
  for (int64_t i = nb * RB; i < nrw; ++i) take(*reinterpret_cast<const Vec *>(p + i * rstep), (I)(first + step * i));
  if (nrw > 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) bi[j] = bi[j] < 0 ? (I)first : bi[j];
  }
  auto merge = [&](const Vec &ov, const IVec &oi) {   // (value, row) pairs of another wave / band into this lane's
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const md_argpair<T> a = RArg<IsMax>::combine(md_argpair<T>{bv[j], bi[j] < 0 ? INT64_MAX : (int64_t)bi[j]},
                                                  md_argpair<T>{ov.v[j], oi.v[j] < 0 ? INT64_MAX : (int64_t)oi.v[j]});
      bv[j] = a.v;
      bi[j] = a.i == INT64_MAX ? (I)-1 : (I)a.i;
    }
  };
  auto pack = [&](Vec &ov, IVec &oi) {
#pragma unroll
    for (int j = 0; j < V; ++j) { ov.v[j] = bv[j]; oi.v[j] = bi[j]; }
  };
  if (w > 0) pack(sv[w - 1][lane], si[w - 1][lane]);
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) merge(sv[k][lane], si[k][lane]);
  }
  auto store_out = [&]() {
#pragma unroll
    for (int j = 0; j < V; ++j) out[col + j] = bi[j] < 0 ? INT64_MAX : (int64_t)bi[j];
  };
  if (NB == 1) {
    if (w == 0 && col_ok) store_out();
    return;
  }
  I *pidx = (I *)pidx_;
  const __amdgpu_buffer_rsrc_t prv = md_rsrc(pval, (unsigned)((int64_t)NB * n_out * (int64_t)sizeof(T)));
  const __amdgpu_buffer_rsrc_t pri = md_rsrc(pidx, (unsigned)((int64_t)NB * n_out * (int64_t)sizeof(I)));
  if (w == 0 && col_ok) {
    Vec ov;
    IVec oi;
    pack(ov, oi);
    md_st16_sc1(prv, (unsigned)(((int64_t)b * n_out + col) * (int64_t)sizeof(T)), ov);
    md_st16_sc1(pri, (unsigned)(((int64_t)b * n_out + col) * (int64_t)sizeof(I)), oi);
  }
  if (!md_ticket_last(tickets + s * MD_TICKET_PAD, (unsigned)NB, &last_flag)) return;
  // the strip's NB partial rows: wave w takes bands w, w + 4, .. in increasing order; then the waves in wave order
#pragma unroll
  for (int j = 0; j < V; ++j) { bv[j] = T(); bi[j] = (I)-1; }
  for (int r = w; r < NB; r += 4) {
    const Vec ov = md_ld16_sc1<Vec>(prv, (unsigned)(((int64_t)r * n_out + col) * (int64_t)sizeof(T)));
    const IVec oi = md_ld16_sc1<IVec>(pri, (unsigned)(((int64_t)r * n_out + col) * (int64_t)sizeof(I)));
    merge(ov, oi);
  }
  __syncthreads();
  if (w > 0) pack(sv[w - 1][lane], si[w - 1][lane]);
  __syncthreads();
  if (w == 0 && col_ok) {
#pragma unroll
    for (int k = 0; k < 3; ++k) merge(sv[k][lane], si[k][lane]);
    store_out();
  }
}

// any / all of a whole contiguous array (a NaN guard: any(isnan(x)); all(mask)): truth values straight from 16-B vectors of the
// array's own type — the general path reads one byte-accumulator element per lane and load (33 MB of bool: 40 us) — block results
// through the two-level ticket of k_reduce_all. ANY: some element != 0 (NaN counts, -0.0 does not); ALL: every element != 0.
template <bool ANY, class T>
__global__ void __launch_bounds__(MD_BLOCK) k_anyall_flat(const T *__restrict__ x, int64_t n, unsigned *partial, unsigned *tickets, uint8_t *out) {
  constexpr int V = 16 / sizeof(T);
  typedef MdVec<T, V> Vec;
  __shared__ unsigned sh[MD_BLOCK / 64];
  __shared__ unsigned last_flag;
  const int64_t nvec = n / V, gs = (int64_t)gridDim.x * blockDim.x;
  const Vec *pv = reinterpret_cast<const Vec *>(x);
  bool r = !ANY;   // ANY: found a true; ALL: still all true
  auto eat = [&](const Vec &t) {
    if constexpr (sizeof(T) == 1) {
      uint32_t w[4];
      __builtin_memcpy(w, &t, 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ANY) r = r | (w[k] != 0u);
        else r = r & ((((w[k] - 0x01010101u) & ~w[k]) & 0x80808080u) == 0u);   // no zero byte in the word
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const bool tv = t.v[j] != (T)0;
        r = ANY ? (r | tv) : (r & tv);
      }
    }
  };
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * gs < nvec; i += 4 * gs) {
    const Vec a = pv[i], b = pv[i + gs], c = pv[i + 2 * gs], d = pv[i + 3 * gs];
    eat(a); eat(b); eat(c); eat(d);
  }
  for (; i < nvec; i += gs) eat(pv[i]);
  for (int64_t k = nvec * V + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gs) {
    const bool tv = x[k] != (T)0;
    r = ANY ? (r | tv) : (r & tv);
  }
  // block verdict: ballots per wave, LDS across waves
  const unsigned long long bal = __ballot(r);
  const unsigned wv = ANY ? (bal != 0ull) : (bal == __ballot(true));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = wv;
  __syncthreads();
  unsigned bv = ANY ? 0u : 1u;
  for (int k = 0; k < MD_BLOCK / 64; ++k) bv = ANY ? (bv | sh[k]) : (bv & sh[k]);
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) out[0] = (uint8_t)bv;
    return;
  }
  if (threadIdx.x == 0) md_st_sc1(partial + blockIdx.x, bv);
  if (!md_ticket_last2(tickets, blockIdx.x, gridDim.x, &last_flag)) return;
  unsigned a = ANY ? 0u : 1u;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += blockDim.x) {
    const unsigned v = md_ld_sc1(partial + k);
    a = ANY ? (a | v) : (a & v);
  }
  const unsigned long long bal2 = __ballot(a != 0u);
  const unsigned wv2 = ANY ? (bal2 != 0ull) : (bal2 == __ballot(true));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = wv2;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned f = ANY ? 0u : 1u;
    for (int k = 0; k < MD_BLOCK / 64; ++k) f = ANY ? (f | sh[k]) : (f & sh[k]);
    out[0] = (uint8_t)f;
  }
}

static int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <bool ANY, class T> static int anyall_flat(const mdhip_array *x, int64_t n, const mdhip_array *out) {
  int64_t blocks = ceil_div(n * (int64_t)sizeof(T), 64 * 1024);   // >= 64 KiB per block
  if (blocks > 4 * MD_NUM_CUS) blocks = 4 * MD_NUM_CUS;
  if (blocks < 1) blocks = 1;
  void *partial = nullptr;
  if (blocks > 1) MD_TRY(mdhip_alloc((size_t)blocks * sizeof(unsigned), &partial));
  MD_LAUNCH((k_anyall_flat<ANY, T>), (unsigned)blocks, MD_BLOCK, (const T *)x->data, n, (unsigned *)partial, md_tickets(), (uint8_t *)out->data);
  const int rc = MD_LAUNCH_CHECK("reduce(any/all, flat)");
  if (partial) mdhip_free(partial);
  return rc;
}

struct HipExec {
  template <class R, class Tacc, class To>
  static int reduce(const MdRedPlan &pl, const mdhip_array *x, const mdhip_array *out) {
    hipStream_t st = md_stream();
    const int64_t n_out = pl.n_out, n_red = pl.n_red;
    if constexpr (md_same<R, RAny>::value || md_same<R, RAll>::value) {
      // the whole of a contiguous array: truth values from 16-B vectors of the array's own type
      if (n_out == 1 && pl.nr == 1 && pl.rx[0] == 1 && n_red >= (1 << 16) && ((uintptr_t)x->data & 15) == 0) {
        constexpr bool ANY = md_same<R, RAny>::value;
        switch (x->dtype) {
          case MDHIP_BOOL: return anyall_flat<ANY, uint8_t>(x, n_red, out);
          case MDHIP_I32: return anyall_flat<ANY, int32_t>(x, n_red, out);
          case MDHIP_I64: return anyall_flat<ANY, int64_t>(x, n_red, out);
          case MDHIP_F32: return anyall_flat<ANY, float>(x, n_red, out);
          case MDHIP_F64: return anyall_flat<ANY, double>(x, n_red, out);
          default: break;
        }
      }
    }
    const bool rows_ok = n_red >= 256 && n_out < (1ll << 30);
    const bool cols_ok = pl.nk >= 1 && pl.kx[pl.nk - 1] == 1 && n_out >= 64;
    if constexpr (sizeof(Tacc) >= 4) {
      // many short contiguous rows: a wave per output (every row must start on a 16-B boundary)
      constexpr int V = 16 / sizeof(Tacc);
      const bool wave_on = md_opt(MD_OPT_ROWS_WAVE) != 0;   // 0: block per output (A/B)
      bool aligned = ((uintptr_t)x->data & 15) == 0;
      for (int k = 0; k < pl.nk; ++k) aligned = aligned && (pl.kx[k] % V) == 0;
      if (wave_on && pl.nr == 1 && pl.rx[0] == 1 && x->dtype == md_dtype_of<Tacc>::value && aligned && n_red >= 32 && n_red <= 64 * V * 8 &&
          n_out >= 1024 && n_out < (1ll << 32)) {
        const unsigned grid = (unsigned)ceil_div(n_out, 4);
        const Tacc *xp = (const Tacc *)x->data;
        const int64_t nvec = n_red / V;
        if (nvec <= 64) MD_LAUNCH((k_reduce_rows_wave<R, Tacc, To, 1>), grid, MD_BLOCK, pl, xp, (To *)out->data);
        else if (nvec <= 128) MD_LAUNCH((k_reduce_rows_wave<R, Tacc, To, 2>), grid, MD_BLOCK, pl, xp, (To *)out->data);
        else if (nvec <= 256) MD_LAUNCH((k_reduce_rows_wave<R, Tacc, To, 4>), grid, MD_BLOCK, pl, xp, (To *)out->data);
        else MD_LAUNCH((k_reduce_rows_wave<R, Tacc, To, 8>), grid, MD_BLOCK, pl, xp, (To *)out->data);
        return MD_LAUNCH_CHECK("reduce(rows,wave)");
      }
    }
    // (rows of the accumulator's own 4- / 8-byte type from 256 elements on keep the block kernel's peeled 16-B loads)
    const bool typed_vec = sizeof(Tacc) >= 4 && x->dtype == md_dtype_of<Tacc>::value;
    if (pl.nr == 1 && pl.rx[0] == 1 && n_red >= 16 && n_red <= 4096 && (!typed_vec || n_red < 256) && n_out >= 256 && md_opt(MD_OPT_ROWS_WAVE) != 0) {
      int64_t blocks = ceil_div(n_out, MD_BLOCK / 64);
      if (blocks > 8 * MD_NUM_CUS) blocks = 8 * MD_NUM_CUS;
      MD_LAUNCH((k_reduce_rows_wave_any<R, Tacc, To>), (unsigned)blocks, MD_BLOCK, pl, x->data, x->dtype, (To *)out->data);
      return MD_LAUNCH_CHECK("reduce(rows,wave,any type)");
    }
    // (kept last axis contiguous, >= 64 outputs: the column kernels, whatever the width — up to round 4 widths under 1024 went to a
    // block per OUTPUT walking its column with the row stride: the bias gradient of a 300,000 x 1000 batch ran at 550 GB/s)
    // (under 1024 outputs only when the contiguous kept axis is at least a wave wide: the middle axis of 8 x 62500 x 8 has 64 outputs in
    // runs of 8 — one block, a quarter of its lanes, 0.24 ms; a block per output walks them in 0.03)
    const bool wide = pl.nk >= 1 && pl.kshape[pl.nk - 1] >= 64;
    if (cols_ok && (!rows_ok || n_out >= 1024 || (wide && (sizeof(Tacc) >= 4 || n_out >= 256)))) {   // (1-byte accumulators — any / all — from 256 columns on)
      if constexpr (sizeof(Tacc) >= 4 && md_same<Tacc, To>::value) {
        constexpr int V = 16 / sizeof(Tacc);
        const bool vec_ok = pl.nk == 1 && pl.nr == 1 && pl.ko[0] == 1 && x->dtype == md_dtype_of<Tacc>::value &&
                            (n_out % V) == 0 && (pl.rx[0] % V) == 0 && ((uintptr_t)x->data & 15) == 0 &&
                            ((uintptr_t)out->data & 15) == 0 && n_red >= 16;
        constexpr int sweep_mode = 1;   // (the tiled kernels below serve what the strips form does not cover: narrow / ragged / short problems)
        const int nb_force = (int)md_opt(MD_OPT_COLS_NB);
        // (f64 / integer max and min keep the tiled kernel — their compare chain wants more waves per CU)
        constexpr bool cheap = md_same<R, RSum>::value || md_same<R, RProd>::value ||
                               ((md_same<R, RMax>::value || md_same<R, RMin>::value) && md_same<Tacc, float>::value);
        constexpr int RB = 8;
        const int64_t NS = ceil_div(n_out, 64 * V);
        // one block per CU (two per CU ran 15 % slower, half a block per CU 20 %: profiles/r3_reduce_lab.txt); <= 64 bands: the
        // last block of a strip holds all its partial rows in registers
        int64_t NB = nb_force > 0 ? nb_force : (NS >= MD_NUM_CUS ? 1 : MD_NUM_CUS / NS);
        if (NB > 64) NB = 64;
        if (NB > n_red / (4 * RB)) NB = n_red / (4 * RB);
        if (NB < 1) NB = 1;
        if (cheap && vec_ok && sweep_mode && n_red >= 512 && NS * NB >= MD_NUM_CUS / 4 && NS * NB < (1ll << 31) &&
            (NB == 1 || NS * MD_TICKET_PAD <= MD_TICKET_WORDS)) {
          void *partial = nullptr;
          if (NB > 1) MD_TRY(mdhip_alloc((size_t)(NB * n_out) * sizeof(Tacc), &partial));
          const Tacc *xp = (const Tacc *)x->data;
          if (n_red * n_out * (int64_t)sizeof(Tacc) > ((int64_t)320 << 20))
            MD_LAUNCH((k_reduce_cols_strips<R, Tacc, RB, true>), (unsigned)(NS * NB), MD_BLOCK, xp, n_out, n_red, pl.rx[0], (int)NS, (int)NB, (Tacc *)partial, md_tickets(), (Tacc *)out->data, (int64_t)0, (int64_t)0);
          else
            MD_LAUNCH((k_reduce_cols_strips<R, Tacc, RB, false>), (unsigned)(NS * NB), MD_BLOCK, xp, n_out, n_red, pl.rx[0], (int)NS, (int)NB, (Tacc *)partial, md_tickets(), (Tacc *)out->data, (int64_t)0, (int64_t)0);
          int rc = MD_LAUNCH_CHECK("reduce(cols,strips)");
          if (partial) mdhip_free(partial);  // stream-ordered: the next user of this block runs after the kernel
          return rc;
        }
        // a middle axis reduced: (outer, n_red, inner) with the inner axis contiguous — `outer` independent column problems in one launch
        if constexpr (cheap) {
          if (sweep_mode && pl.nk == 2 && pl.nr == 1 && pl.kx[1] == 1 && pl.ko[1] == 1 && pl.ko[0] == pl.kshape[1] && x->dtype == md_dtype_of<Tacc>::value &&
              (pl.kshape[1] % V) == 0 && (pl.rx[0] % V) == 0 && (pl.kx[0] % V) == 0 && ((uintptr_t)x->data & 15) == 0 && ((uintptr_t)out->data & 15) == 0 &&
              pl.kshape[0] <= 65535 && pl.kshape[1] >= 256 && n_red >= 64) {
            const int64_t outer = pl.kshape[0], inner = pl.kshape[1];
            const int64_t NSb = ceil_div(inner, 64 * V);
            int64_t NBb = NSb * outer >= MD_NUM_CUS ? 1 : MD_NUM_CUS / (NSb * outer);
            if (NBb > 64) NBb = 64;
            if (NBb > n_red / (4 * RB)) NBb = n_red / (4 * RB);
            if (NBb < 1) NBb = 1;
            if (NBb == 1 || NSb * outer * MD_TICKET_PAD <= MD_TICKET_WORDS) {
              void *partial = nullptr;
              if (NBb > 1) MD_TRY(mdhip_alloc((size_t)(outer * NBb * inner) * sizeof(Tacc), &partial));
              const dim3 grid((unsigned)(NSb * NBb), (unsigned)outer);
              MD_LAUNCH((k_reduce_cols_strips<R, Tacc, RB, false>), grid, MD_BLOCK, (const Tacc *)x->data, inner, n_red, pl.rx[0], (int)NSb, (int)NBb, (Tacc *)partial,
                        md_tickets(), (Tacc *)out->data, (int64_t)pl.kx[0], (int64_t)inner);
              int rc = MD_LAUNCH_CHECK("reduce(cols,strips,batched)");
              if (partial) mdhip_free(partial);
              return rc;
            }
          }
        }
        if (vec_ok) {
          const int64_t bxv = ceil_div(n_out, 64 * V);
          int64_t splits = 1024 / bxv;
          if (splits > n_red / 64) splits = n_red / 64;
          if (splits > 65535) splits = 65535;
          if (splits < 1) splits = 1;
          const int64_t chunk = ceil_div(ceil_div(n_red, splits), 16) * 16;
          splits = ceil_div(n_red, chunk);
          const Tacc *xp = (const Tacc *)x->data;
          if (splits == 1) {
            k_reduce_cols_vec<R, Tacc, To, true><<<dim3((unsigned)bxv, 1), MD_BLOCK, 0, st>>>(xp, n_out, n_red, pl.rx[0], chunk, (To *)out->data);
            return MD_LAUNCH_CHECK("reduce(cols,vec)");
          }
          void *partial = nullptr;
          MD_TRY(mdhip_alloc((size_t)(splits * n_out) * sizeof(Tacc), &partial));
          k_reduce_cols_vec<R, Tacc, Tacc, false><<<dim3((unsigned)bxv, (unsigned)splits), MD_BLOCK, 0, st>>>(xp, n_out, n_red, pl.rx[0], chunk, (Tacc *)partial);
          const int64_t chunk2 = ceil_div(splits, 16) * 16;
          k_reduce_cols_vec<R, Tacc, To, true><<<dim3((unsigned)bxv, 1), MD_BLOCK, 0, st>>>((const Tacc *)partial, n_out, splits, n_out, chunk2, (To *)out->data);
          int rc = MD_LAUNCH_CHECK("reduce(cols,vec,split)");
          mdhip_free(partial);
          return rc;
        }
      }
      const int64_t bx = ceil_div(n_out, MD_BLOCK);
      int64_t splits = 1024 / bx;
      if (splits > n_red / 32) splits = n_red / 32;
      if (splits > 65535) splits = 65535;
      if (splits < 1) splits = 1;
      const int64_t chunk = ceil_div(n_red > 0 ? n_red : 1, splits);
      splits = ceil_div(n_red > 0 ? n_red : 1, chunk);
      dim3 grid((unsigned)bx, (unsigned)splits);
      if (splits == 1) {
        k_reduce_cols<R, Tacc, To, true><<<grid, MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, chunk, (To *)out->data);
        return MD_LAUNCH_CHECK("reduce(cols)");
      }
      void *partial = nullptr;
      MD_TRY(mdhip_alloc((size_t)(splits * n_out) * sizeof(Tacc), &partial));
      k_reduce_cols<R, Tacc, Tacc, false><<<grid, MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, chunk, (Tacc *)partial);
      k_finish_cols<R, Tacc, To><<<(unsigned)bx, MD_BLOCK, 0, st>>>(pl, (const Tacc *)partial, splits, (To *)out->data);
      int rc = MD_LAUNCH_CHECK("reduce(cols,split)");
      mdhip_free(partial);  // stream-ordered: the next user of this block runs after the finish pass
      return rc;
    }
    if (rows_ok) {
      // ~1024 blocks in total (4 per CU; option rows_blocks overrides: experiments); each block should still see >= 4096 items
      const int64_t total_blocks = md_opt(MD_OPT_ROWS_BLOCKS) > 0 ? md_opt(MD_OPT_ROWS_BLOCKS) : 1024;
      int64_t splits = total_blocks / n_out;
      const int64_t max_splits = ceil_div(n_red, 4096);
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
      if (splits == 1) {
        k_reduce_rows<R, Tacc, To, 0><<<(unsigned)n_out, MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, 1, (To *)out->data, nullptr, nullptr);
        return MD_LAUNCH_CHECK("reduce(rows)");
      }
      void *partial = nullptr;
      MD_TRY(mdhip_alloc((size_t)(splits * n_out) * sizeof(Tacc), &partial));
      const bool nt = n_red * n_out * (int64_t)sizeof(Tacc) > ((int64_t)320 << 20);
      int rc;
      if constexpr (sizeof(Tacc) >= 4) {
        constexpr bool all_on = true;
        // the whole of a contiguous array of a STORAGE-ONLY dtype (sum(int8), max(float16) ..): the same kernel, 16-B loads of the narrow type
        if (all_on && n_out == 1 && pl.nr == 1 && pl.rx[0] == 1 && md_is_narrow(x->dtype) && ((uintptr_t)x->data & 15) == 0) {
#define MD_RED_ALL_AS(DT, S)                                                                                                          \
          if (x->dtype == DT) {                                                                                                        \
            if (nt) MD_LAUNCH((k_reduce_all<R, Tacc, To, true, S>), (unsigned)splits, MD_BLOCK, (const S *)x->data, n_red, (Tacc *)partial, md_tickets(), (To *)out->data); \
            else MD_LAUNCH((k_reduce_all<R, Tacc, To, false, S>), (unsigned)splits, MD_BLOCK, (const S *)x->data, n_red, (Tacc *)partial, md_tickets(), (To *)out->data);   \
            rc = MD_LAUNCH_CHECK("reduce(all, storage-only dtype)");                                                                   \
            mdhip_free(partial);                                                                                                       \
            return rc;                                                                                                                 \
          }
          // (only the accumulator types a storage-only input can meet: md_dispatch.h)
          if constexpr (md_same<Tacc, int64_t>::value && (md_same<R, RSum>::value || md_same<R, RProd>::value)) {
            MD_RED_ALL_AS(MDHIP_I8, int8_t) MD_RED_ALL_AS(MDHIP_I16, int16_t) MD_RED_ALL_AS(MDHIP_U8, uint8_t) MD_RED_ALL_AS(MDHIP_U16, uint16_t)
            MD_RED_ALL_AS(MDHIP_U32, uint32_t) MD_RED_ALL_AS(MDHIP_U64, uint64_t)
          }
          if constexpr (md_same<Tacc, int32_t>::value && (md_same<R, RMax>::value || md_same<R, RMin>::value)) {
            MD_RED_ALL_AS(MDHIP_I8, int8_t) MD_RED_ALL_AS(MDHIP_I16, int16_t) MD_RED_ALL_AS(MDHIP_U8, uint8_t) MD_RED_ALL_AS(MDHIP_U16, uint16_t)
          }
          if constexpr (md_same<Tacc, int64_t>::value && (md_same<R, RMax>::value || md_same<R, RMin>::value)) { MD_RED_ALL_AS(MDHIP_U32, uint32_t) }
          if constexpr (md_same<Tacc, float>::value) { MD_RED_ALL_AS(MDHIP_F16, f16) }
#undef MD_RED_ALL_AS
        }
        if (all_on && n_out == 1 && pl.nr == 1 && pl.rx[0] == 1 && x->dtype == md_dtype_of<Tacc>::value && ((uintptr_t)x->data & 15) == 0) {
          if (nt) MD_LAUNCH((k_reduce_all<R, Tacc, To, true>), (unsigned)splits, MD_BLOCK, (const Tacc *)x->data, n_red, (Tacc *)partial, md_tickets(), (To *)out->data);
          else MD_LAUNCH((k_reduce_all<R, Tacc, To, false>), (unsigned)splits, MD_BLOCK, (const Tacc *)x->data, n_red, (Tacc *)partial, md_tickets(), (To *)out->data);
          rc = MD_LAUNCH_CHECK("reduce(all)");
          mdhip_free(partial);
          return rc;
        }
        constexpr bool ticket_on = true;   // (the two-launch form below: more outputs x splits than ticket words, 1-byte accumulators)
        if (ticket_on && n_out * (splits >= 64 ? MD_TICKET2_WORDS : MD_TICKET_PAD) <= MD_TICKET_WORDS) {
          if (nt) MD_LAUNCH((k_reduce_rows<R, Tacc, To, 2, true>), (unsigned)(n_out * splits), MD_BLOCK, pl, x->data, x->dtype, splits, (To *)out->data, (Tacc *)partial, md_tickets());
          else MD_LAUNCH((k_reduce_rows<R, Tacc, To, 2>), (unsigned)(n_out * splits), MD_BLOCK, pl, x->data, x->dtype, splits, (To *)out->data, (Tacc *)partial, md_tickets());
          rc = MD_LAUNCH_CHECK("reduce(rows,ticket)");
          mdhip_free(partial);
          return rc;
        }
      }
      if (nt) k_reduce_rows<R, Tacc, Tacc, 1, true><<<(unsigned)(n_out * splits), MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, splits, (Tacc *)partial, nullptr, nullptr);
      else k_reduce_rows<R, Tacc, Tacc, 1><<<(unsigned)(n_out * splits), MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, splits, (Tacc *)partial, nullptr, nullptr);
      k_finish_rows<R, Tacc, To><<<(unsigned)n_out, MD_BLOCK, 0, st>>>(pl, (const Tacc *)partial, splits, (To *)out->data);
      rc = MD_LAUNCH_CHECK("reduce(rows,split)");
      mdhip_free(partial);
      return rc;
    }
    k_reduce_generic<R, Tacc, To><<<md_grid_for(n_out), MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, (To *)out->data);
    return MD_LAUNCH_CHECK("reduce(generic)");
  }

  template <bool IsMax, class T>
  static int argreduce(const MdRedPlan &pl, const mdhip_array *x, const mdhip_array *out) {
    hipStream_t st = md_stream();
    if constexpr (sizeof(T) >= 4) {
      constexpr int V = 16 / sizeof(T);
      const bool same = x->dtype == md_dtype_of<T>::value;
      // kept axis contiguous (argmax over the rows of a row-major matrix)
      if (same && pl.nk == 1 && pl.nr == 1 && pl.kx[0] == 1 && pl.ko[0] == 1 && pl.n_out >= 256 && (pl.n_out % V) == 0 &&
          (pl.rx[0] % V) == 0 && ((uintptr_t)x->data & 15) == 0 && pl.n_red >= 16) {
        const int64_t n_out = pl.n_out, n_red = pl.n_red;
        const int64_t bxv = ceil_div(n_out, 64 * V);
        constexpr bool strips_on = true;   // (the chunked kernel + finish pass below: more strips than ticket words, short columns)
        if (strips_on && n_red < (1ll << 31) && n_red >= 64 && bxv * MD_TICKET_PAD <= MD_TICKET_WORDS) {
          const int64_t NS = bxv;
          const int arg_blocks = md_opt(MD_OPT_ARG_BLOCKS) > 0 ? (int)md_opt(MD_OPT_ARG_BLOCKS) : MD_NUM_CUS;   // one block per CU, as the column sums (rocprofv3: 26.6 us against 38.0 with four per CU); the knob is for experiments
          int64_t NB = ceil_div(arg_blocks, NS);
          if (NB > 64) NB = 64;
          if (NB > n_red / 32) NB = n_red / 32;
          if (NB < 1) NB = 1;
          while (NB > 1 && NB * n_out * 8 >= (1ll << 31)) NB /= 2;   // (32-bit byte offsets into the partial rows)
          constexpr int ARG_RB = 4;   // (rocprofv3 at one block per CU: 26.6 us with 4 rows per batch, 28.4 with 8)
          void *pv = nullptr, *pi = nullptr;
          if (NB > 1) {
            MD_TRY(mdhip_alloc((size_t)(NB * n_out) * sizeof(T), &pv));
            const int rc = mdhip_alloc((size_t)(NB * n_out) * (V == 4 ? 4 : 8), &pi);
            if (rc != MDHIP_OK) { mdhip_free(pv); return rc; }
          }
          MD_LAUNCH((k_arg_cols_strips<IsMax, T, ARG_RB>), (unsigned)(NS * NB), MD_BLOCK, (const T *)x->data, n_out, n_red, pl.rx[0], (int)NS, (int)NB, (T *)pv, pi, md_tickets(),
                    (int64_t *)out->data);
          const int rc = MD_LAUNCH_CHECK("argreduce(cols,strips)");
          if (pv) mdhip_free(pv);
          if (pi) mdhip_free(pi);
          return rc;
        }
        int64_t splits = 1024 / bxv;
        if (splits > n_red / 64) splits = n_red / 64;
        if (splits > 65535) splits = 65535;
        if (splits < 1) splits = 1;
        const int64_t chunk = ceil_div(ceil_div(n_red, splits), 16) * 16;
        splits = ceil_div(n_red, chunk);
        const T *xp = (const T *)x->data;
        if (splits == 1) {
          k_arg_cols_vec<IsMax, T, true><<<dim3((unsigned)bxv, 1), MD_BLOCK, 0, st>>>(xp, n_out, n_red, pl.rx[0], chunk, nullptr, (int64_t *)out->data);
          return MD_LAUNCH_CHECK("argreduce(cols,vec)");
        }
        void *pv = nullptr, *pi = nullptr;
        MD_TRY(mdhip_alloc((size_t)(splits * n_out) * sizeof(T), &pv));
        int rc = mdhip_alloc((size_t)(splits * n_out) * sizeof(int64_t), &pi);
        if (rc != MDHIP_OK) { mdhip_free(pv); return rc; }
        k_arg_cols_vec<IsMax, T, false><<<dim3((unsigned)bxv, (unsigned)splits), MD_BLOCK, 0, st>>>(xp, n_out, n_red, pl.rx[0], chunk, (T *)pv, (int64_t *)pi);
        k_arg_cols_finish<IsMax, T><<<(unsigned)ceil_div(n_out, MD_BLOCK), MD_BLOCK, 0, st>>>((const T *)pv, (const int64_t *)pi, n_out, splits, (int64_t *)out->data);
        rc = MD_LAUNCH_CHECK("argreduce(cols,vec,split)");
        mdhip_free(pv);
        mdhip_free(pi);
        return rc;
      }
      // reduced axis contiguous
      if (same && pl.nr == 1 && pl.rx[0] == 1 && pl.n_red >= 1024 && pl.n_out < (1ll << 30)) {
        // ~2048 blocks in total, each with >= 8192 items of its row
        int64_t splits = 2048 / pl.n_out;
        const int64_t max_splits = ceil_div(pl.n_red, 8192);
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        const int64_t chunk = ceil_div(ceil_div(pl.n_red, splits), V) * V;
        splits = ceil_div(pl.n_red, chunk);
        if (splits == 1) {
          k_arg_rows_vec<IsMax, T, true><<<(unsigned)pl.n_out, MD_BLOCK, 0, st>>>(pl, (const T *)x->data, 1, chunk, nullptr, (int64_t *)out->data);
          return MD_LAUNCH_CHECK("argreduce(rows,vec)");
        }
        void *pv = nullptr, *pi = nullptr;
        MD_TRY(mdhip_alloc((size_t)(splits * pl.n_out) * sizeof(T), &pv));
        int rc = mdhip_alloc((size_t)(splits * pl.n_out) * sizeof(int64_t), &pi);
        if (rc != MDHIP_OK) { mdhip_free(pv); return rc; }
        k_arg_rows_vec<IsMax, T, false><<<(unsigned)(pl.n_out * splits), MD_BLOCK, 0, st>>>(pl, (const T *)x->data, splits, chunk, (T *)pv, (int64_t *)pi);
        k_arg_rows_finish<IsMax, T><<<(unsigned)ceil_div(pl.n_out, MD_BLOCK), MD_BLOCK, 0, st>>>(pl, (const T *)pv, (const int64_t *)pi, splits, (int64_t *)out->data);
        rc = MD_LAUNCH_CHECK("argreduce(rows,vec,split)");
        mdhip_free(pv);
        mdhip_free(pi);
        return rc;
      }
    }
    const bool typed = x->dtype == md_dtype_of<T>::value;
    if (pl.nr == 1 && pl.rx[0] == 1 && pl.n_red >= 24 && (pl.n_red < 1024 || !typed) && pl.n_red <= 65536 && pl.n_out >= 64) {
      int64_t blocks = ceil_div(pl.n_out, MD_BLOCK / 64);
      if (blocks > 8 * MD_NUM_CUS) blocks = 8 * MD_NUM_CUS;
      if (typed) MD_LAUNCH((k_arg_rows_wave<IsMax, T, true>), (unsigned)blocks, MD_BLOCK, pl, x->data, x->dtype, (int64_t *)out->data);
      else MD_LAUNCH((k_arg_rows_wave<IsMax, T, false>), (unsigned)blocks, MD_BLOCK, pl, x->data, x->dtype, (int64_t *)out->data);
      return MD_LAUNCH_CHECK("argreduce(rows,wave)");
    }
    if (pl.n_red >= 512 && pl.n_out < (1ll << 30)) {
      k_arg_block<IsMax, T><<<(unsigned)pl.n_out, MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, (int64_t *)out->data);
      return MD_LAUNCH_CHECK("argreduce(block)");
    }
    k_arg_thread<IsMax, T><<<md_grid_for(pl.n_out), MD_BLOCK, 0, st>>>(pl, x->data, x->dtype, (int64_t *)out->data);
    return MD_LAUNCH_CHECK("argreduce(thread)");
  }
};

}  // namespace

extern "C" int mdhip_reduce(int op, const mdhip_array *x, const mdhip_array *out, uint32_t mask) {
  return md_reduce_any_out<HipExec>(op, x, out, mask);   // (storage-only input dtypes load through md_load; storage-only results: md_narrow.h)
}
