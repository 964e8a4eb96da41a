// skinny.hip — matrix products with one THIN side: matrix x vector, vector x matrix, up to eight columns / rows, and the
// bottom / right strips of a peeled ragged product (gemm.hip, launch_mfma_peeled).
//
// Serves reference minidiff/backend/numpy.py:84 (np.matmul) as called by minidiff/ops/definitions.py:487-492 when an operand is a
// vector or a few columns wide. Such a product is HBM-bound — 2 flop per 4-byte element of the big operand, which is read exactly
// once — so the matrix cores have nothing to offer: on the 64x64 MFMA tiles + split-K these shapes ran at 22-30 % of the 8 TB/s
// peak (8192^2 x vector: 111-150 us for 256 MiB). Two streaming kernels instead, both computing
//
//     out[r][c] = sum_k X[r][k] * Y[k][c]        r < R (large), c < NC <= 8
//
// for whichever operand is the big one (a thin-M product is the thin-N product of the transposed problem: X = B^T, Y = A^T, out = C^T,
// all by strides):
//   k_skinny_rowdot   X unit-stride along k: a wave owns RW rows, lanes stride k with 16-B loads (1 KiB per wave and row), the
//                     thin operand's k-chunk sits in LDS as [c][k]; lane partials -> wave shuffle tree -> one store per output.
//   k_skinny_colsum   X unit-stride along r: a weighted column sum — a lane owns V neighbouring r, the block's four waves take
//                     interleaved k rows of one of NB bands, Y[k][c] arrive as wave-uniform (scalar) loads; band partials are
//                     published write-through and the block that arrives LAST at the strip's ticket adds them in band order
//                     (md_ticket.h): one launch, fixed order, bit-identical run to run.
// Algorithmic bytes: R * K * sizeof(T) per launch (the thin operand and the result are noise). float32 and float64.
#include "md_hip.h"

extern "C" int mdhip_alloc(size_t, void **);
extern "C" int mdhip_free(void *);

namespace {

struct SkinnyArgs {
  const void *X, *Y;
  void *out;
  int64_t R, K;
  int64_t xr, xk, x_bs;   // element strides of X[r][k] (one of xr / xk is 1), batch stride
  int64_t yk, yc, y_bs;   // Y[k][c]
  int64_t o_r, o_c, o_bs; // out[r][c]
  int nc;                 // live columns (<= NCT)
};

template <class T> __device__ __forceinline__ T md_readlane(T v, int l) {   // l: compile-time constant after unrolling
  if constexpr (sizeof(T) == 8) {
    union { T t; int w[2]; } u;
    u.t = v;
    u.w[0] = __builtin_amdgcn_readlane(u.w[0], l);
    u.w[1] = __builtin_amdgcn_readlane(u.w[1], l);
    return u.t;
  } else {
    union { T t; int w; } u;
    u.t = v;
    u.w = __builtin_amdgcn_readlane(u.w, l);
    return u.t;
  }
}

// ------------------------------------------------------------------------------------------------ X k-contiguous
// The thin operand goes through LDS in k-chunks of KC elements, [c][k] so that a lane reads its V consecutive k of one column as one
// 16-B ds_read. Two buffers: the global loads of chunk i + 1 are issued (into registers) before chunk i is multiplied and stored
// behind it, so their latency hides under the X stream; one barrier per chunk.
template <class T, int NCT, int RW>
__global__ void __launch_bounds__(256) k_skinny_rowdot(SkinnyArgs a) {
  constexpr int V = 16 / sizeof(T);
  constexpr int KC = 4096 / sizeof(T);          // 4 KiB of k per column and buffer
  constexpr int PER = NCT * KC / 256;           // staged elements per thread and chunk
  constexpr int STEPS = KC / (64 * V);          // 16-B steps of a lane per chunk (4)
  typedef MdVec<T, V> Vec;
  __shared__ __attribute__((aligned(16))) T Ys[2][NCT][KC + V];   // (+ 16 B per row: the stores of one k for all c fall into different banks)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const T *X = (const T *)a.X + (int64_t)blockIdx.z * a.x_bs;
  const T *Y = (const T *)a.Y + (int64_t)blockIdx.z * a.y_bs;
  T *out = (T *)a.out + (int64_t)blockIdx.z * a.o_bs;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + w) * RW;
  const T *xp[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int64_t r = r0 + i < a.R ? r0 + i : a.R - 1;   // rows past the end re-read the last row and store nothing
    xp[i] = X + r * a.xr;
  }
  T acc[RW][NCT];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[i][c] = (T)0;
  const bool y_along_k = a.yk == 1;   // Y rows contiguous along k (a vector, a few k-contiguous rows): threads run along k; else along c
  T stage[PER];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int c = y_along_k ? idx / KC : idx % NCT, k = y_along_k ? idx % KC : idx / NCT;
      stage[u] = (k0 + k < a.K && c < a.nc) ? Y[(k0 + k) * a.yk + c * a.yc] : (T)0;
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int c = y_along_k ? idx / KC : idx % NCT, k = y_along_k ? idx % KC : idx / NCT;
      Ys[buf][c][k] = stage[u];
    }
  };
  fetch(0);
  put(0);
  __syncthreads();
  int buf = 0;
  for (int64_t k0 = 0; k0 < a.K; k0 += KC, buf ^= 1) {
    const bool more = k0 + KC < a.K;
    if (more) fetch(k0 + KC);
    const int kn = (int)(a.K - k0 < KC ? a.K - k0 : KC);
    Vec x[STEPS][RW];
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int k = st * 64 * V + lane * V;
      const int kk = k < kn ? k : 0;     // past the end of the last chunk: a valid address, multiplied by the zeros of the staged Y
#pragma unroll
      for (int i = 0; i < RW; ++i) x[st][i] = *reinterpret_cast<const Vec *>(xp[i] + k0 + kk);
    }
#pragma unroll
    for (int st = 0; st < STEPS; ++st) {
      const int k = st * 64 * V + lane * V;
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const Vec y = *reinterpret_cast<const Vec *>(&Ys[buf][c][k]);
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
          for (int j = 0; j < V; ++j) acc[i][c] = fma(x[st][i].v[j], y.v[j], acc[i][c]);
      }
    }
    if (more) put(buf ^ 1);
    __syncthreads();
  }
  // lane partials -> lane 0, fixed tree
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      T v = acc[i][c];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v += md_shfl_down(v, d);
      acc[i][c] = v;
    }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < RW; ++i)
      if (r0 + i < a.R) {
#pragma unroll
        for (int c = 0; c < NCT; ++c)
          if (c < a.nc) out[(r0 + i) * a.o_r + c * a.o_c] = acc[i][c];
      }
  }
}

// ------------------------------------------------------------------------------------------------ X r-contiguous
template <class T, int NCT, int RB>
__global__ void __launch_bounds__(256) k_skinny_colsum(SkinnyArgs a, int NS, int NB, T *partial, unsigned *tickets) {
  constexpr int V = 16 / sizeof(T);
  typedef MdVec<T, V> Vec;
  __shared__ Vec sm[3][NCT][64];
  __shared__ unsigned last_flag;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x % NS, b = blockIdx.x / NS;
  const T *X = (const T *)a.X + (int64_t)blockIdx.z * a.x_bs;
  const T *Y = (const T *)a.Y + (int64_t)blockIdx.z * a.y_bs;
  T *out = (T *)a.out + (int64_t)blockIdx.z * a.o_bs;
  const int64_t col_raw = ((int64_t)s * 64 + lane) * V;
  const bool col_ok = col_raw < a.R;
  const int64_t col = col_ok ? col_raw : a.R - V;   // lanes past the edge load a valid vector and store nothing (R % V == 0, R >= V)
  const int64_t first = b + (int64_t)NB * w, step = (int64_t)NB * 4;   // this wave's k rows: first + step * i (interleaved across bands and waves)
  const int64_t nrw = first < a.K ? (a.K - first + step - 1) / step : 0;
  const int64_t nb = nrw / RB;
  T acc[NCT][V];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[c][j] = (T)0;
  const T *p = X + col + first * a.xk;
  const int64_t rstep = step * a.xk;
  const T *yq = Y + first * a.yk;          // wave-uniform: the Y reads below are scalar loads
  const int64_t ystep = step * a.yk;
  Vec t[2][RB];
  T ty[2];   // the batch's RB x NCT values of Y, one per lane (lane = u * NCT + c): ONE vector load per batch, broadcast by v_readlane
  static_assert(RB * NCT <= 64, "a batch's Y values must fit one wave-wide load");
  const int yu = lane / NCT, yc_ = lane % NCT;
  const bool y_lane = lane < RB * NCT && yc_ < a.nc;
  auto load = [&](int buf, int64_t bt) {
    const int64_t i0 = (bt < nb ? bt : nb - 1) * RB;   // a prefetch past the end re-reads the last batch (discarded)
#pragma unroll
    for (int u = 0; u < RB; ++u) t[buf][u] = *reinterpret_cast<const Vec *>(p + (i0 + u) * rstep);
    ty[buf] = y_lane ? yq[(i0 + yu) * ystep + yc_ * a.yc] : (T)0;
  };
  auto mac = [&](int buf, int64_t) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const T yv = md_readlane(ty[buf], u * NCT + c);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[c][j] = fma(t[buf][u].v[j], yv, acc[c][j]);
      }
    }
  };
  if (nb > 0) {
    load(0, 0);
    int64_t bt = 0;
    for (; bt + 1 < nb; bt += 2) {
      load(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
      mac(0, bt);
      __builtin_amdgcn_sched_barrier(0);
      load(0, bt + 2);
      __builtin_amdgcn_sched_barrier(0);
      mac(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bt < nb) mac(0, bt);
  }
  for (int64_t i = nb * RB; i < nrw; ++i) {
    const Vec tt = *reinterpret_cast<const Vec *>(p + i * rstep);
    const T *yr = yq + i * ystep;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const T yv = c < a.nc ? yr[c * a.yc] : (T)0;
#pragma unroll
      for (int j = 0; j < V; ++j) acc[c][j] = fma(tt.v[j], yv, acc[c][j]);
    }
  }
  // waves 1..3 -> wave 0, in wave order
  if (w > 0) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int j = 0; j < V; ++j) sm[w - 1][c][lane].v[j] = acc[c][j];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[c][j] += sm[q][c][lane].v[j];
  }
  auto store_out = [&]() {   // out[r][c], r = col + j
#pragma unroll
    for (int c = 0; c < NCT; ++c)
      if (c < a.nc) {
        if (a.o_r == 1) {
          Vec o;
#pragma unroll
          for (int j = 0; j < V; ++j) o.v[j] = acc[c][j];
          T *dst = out + col + c * a.o_c;
          if (((uintptr_t)dst & 15) == 0) { *reinterpret_cast<Vec *>(dst) = o; continue; }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) out[(col + j) * a.o_r + c * a.o_c] = acc[c][j];
      }
  };
  if (NB == 1) {
    if (w == 0 && col_ok) store_out();
    return;
  }
  // partial[z][b][c][r]
  const int64_t plane = (int64_t)NCT * a.R;
  T *pz = partial + (int64_t)blockIdx.z * NB * plane;
  const __amdgpu_buffer_rsrc_t pr = md_rsrc(pz, (unsigned)((int64_t)NB * plane * (int64_t)sizeof(T)));
  if (w == 0 && col_ok) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      Vec o;
#pragma unroll
      for (int j = 0; j < V; ++j) o.v[j] = acc[c][j];
      md_st16_sc1(pr, (unsigned)(((int64_t)b * plane + (int64_t)c * a.R + col) * (int64_t)sizeof(T)), o);
    }
  }
  if (!md_ticket_last(tickets + ((int64_t)blockIdx.z * NS + s) * MD_TICKET_PAD, (unsigned)NB, &last_flag)) return;
  // the strip's NB partial planes, band order: wave w takes bands w, w + 4, .. then the waves combine in wave order
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[c][j] = (T)0;
  for (int r = w; r < NB; r += 4) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const Vec pt = md_ld16_sc1<Vec>(pr, (unsigned)(((int64_t)r * plane + (int64_t)c * a.R + col) * (int64_t)sizeof(T)));
#pragma unroll
      for (int j = 0; j < V; ++j) acc[c][j] += pt.v[j];
    }
  }
  __syncthreads();
  if (w > 0) {
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int j = 0; j < V; ++j) sm[w - 1][c][lane].v[j] = acc[c][j];
  }
  __syncthreads();
  if (w == 0 && col_ok) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[c][j] += sm[q][c][lane].v[j];
    store_out();
  }
}

template <class T, int NCT> int launch_rowdot(const SkinnyArgs &a, int64_t batch) {
  const int rw = a.R >= 8192 ? 4 : a.R >= 4096 ? 2 : 1;
  const dim3 grid((unsigned)((a.R + 4 * rw - 1) / (4 * rw)), 1, (unsigned)batch);
  if (rw == 4) MD_LAUNCH((k_skinny_rowdot<T, NCT, 4>), grid, 256, a);
  else if (rw == 2) MD_LAUNCH((k_skinny_rowdot<T, NCT, 2>), grid, 256, a);
  else MD_LAUNCH((k_skinny_rowdot<T, NCT, 1>), grid, 256, a);
  return MD_LAUNCH_CHECK("matmul(skinny, row dot)");
}

template <class T, int NCT> int launch_colsum(const SkinnyArgs &a, int64_t batch) {
  constexpr int V = 16 / sizeof(T);
  const int64_t NS = (a.R + 64 * V - 1) / (64 * V);
  int64_t NB = (1024 + NS * batch - 1) / (NS * batch);   // ~1024 blocks
  if (NB > 64) NB = 64;
  if (NB > a.K / 32) NB = a.K / 32;
  if (NB < 1) NB = 1;
  if (NB > 1 && NS * batch * MD_TICKET_PAD > MD_TICKET_WORDS) NB = 1;
  while (NB > 1 && NB * NCT * a.R * (int64_t)sizeof(T) >= (1ll << 31)) NB /= 2;   // (32-bit byte offsets into a batch's partial planes)
  void *partial = nullptr;
  if (NB > 1) MD_TRY(mdhip_alloc((size_t)(batch * NB * NCT * a.R) * sizeof(T), &partial));
  const dim3 grid((unsigned)(NS * NB), 1, (unsigned)batch);
  MD_LAUNCH((k_skinny_colsum<T, NCT, 8>), grid, 256, a, (int)NS, (int)NB, (T *)partial, md_tickets());
  int rc = MD_LAUNCH_CHECK("matmul(skinny, weighted column sum)");
  if (partial) mdhip_free(partial);   // stream-ordered
  return rc;
}

template <class T> int skinny_run(const SkinnyArgs &a, int64_t batch, bool rowdot) {
#define MD_SK(NCT) (rowdot ? launch_rowdot<T, NCT>(a, batch) : launch_colsum<T, NCT>(a, batch))
  if (a.nc == 1) return MD_SK(1);
  if (a.nc == 2) return MD_SK(2);
  if (a.nc <= 4) return MD_SK(4);
  return MD_SK(8);
#undef MD_SK
}

}  // namespace

// ---- few outputs, k long: a dot product, (1,K)@(K,1), a few rows times a few columns, (64,K)@(K,64) ----------------------------
// A product with at most a few thousand outputs and k in the millions gives a tile kernel ONE block (or a handful) that walks k
// alone: 27-80 ns per k — a 4-million-element np.dot took 115 ms, (16, 2^20) @ (2^20, 16) 39 ms. Here k is cut over up to ~2048
// blocks: a block owns an (up to) 8 x 8 patch of the output and a k range, a thread strides that range and keeps the patch's
// partial sums, the block folds them (wave shuffles, then LDS), and a second small launch adds the block partials in block order —
// fixed order, bit-identical run to run. float32 / float64 / int32 / int64 (integers wrap as NumPy's).
namespace {
struct LongKArgs {
  const void *A, *B;
  void *C, *partial;
  int64_t K, chunk;
  int64_t a_bs, a_ms, a_ks, b_bs, b_ks, b_ns, c_bs, c_ms, c_ns;
  int M, N, nb, tiles_m, tiles_n;
};
template <class T> __device__ __forceinline__ T longk_mac(T acc, T x, T y) {
  if constexpr (md_is_float<T>::value) return x * y + acc;   // (contracted to an fma)
  else return BAdd::apply(acc, BMul::apply(x, y));            // integers wrap, as NumPy's
}
template <class T, int TM, int TN>
__global__ void __launch_bounds__(MD_BLOCK) k_longk_partial(LongKArgs g) {
  __shared__ T red[MD_BLOCK / 64][TM * TN];
  const int tiles = g.tiles_m * g.tiles_n, bz = blockIdx.y / tiles, tile = blockIdx.y - bz * tiles;
  const int m0 = (tile / g.tiles_n) * TM, n0 = (tile % g.tiles_n) * TN;
  const T *A = (const T *)g.A + (int64_t)bz * g.a_bs + (int64_t)m0 * g.a_ms;
  const T *B = (const T *)g.B + (int64_t)bz * g.b_bs + (int64_t)n0 * g.b_ns;
  const int mm = g.M - m0, nn = g.N - n0;   // live rows / columns of this patch
  const int64_t k0 = (int64_t)blockIdx.x * g.chunk, k1 = k0 + g.chunk < g.K ? k0 + g.chunk : g.K;
  T acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc[m][n] = T(0);
  for (int64_t k = k0 + threadIdx.x; k < k1; k += MD_BLOCK) {
    T a[TM], b[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) a[m] = m < mm ? A[m * g.a_ms + k * g.a_ks] : T(0);
#pragma unroll
    for (int n = 0; n < TN; ++n) b[n] = n < nn ? B[k * g.b_ks + n * g.b_ns] : T(0);
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n) acc[m][n] = longk_mac<T>(acc[m][n], a[m], b[n]);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      T v = acc[m][n];
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) v = BAdd::apply(v, (T)__shfl_down(v, d, 64));
      if (lane == 0) red[w][m * TN + n] = v;
    }
  __syncthreads();
  if (threadIdx.x < TM * TN) {
    T v = red[0][threadIdx.x];
#pragma unroll
    for (int ww = 1; ww < MD_BLOCK / 64; ++ww) v = BAdd::apply(v, red[ww][threadIdx.x]);
    ((T *)g.partial)[((int64_t)blockIdx.y * g.nb + blockIdx.x) * (TM * TN) + threadIdx.x] = v;
  }
}
// one block per patch: the 256 threads split into G = 256 / (TM TN) groups, group q adds the partials of blocks q, q + G, q + 2G, ..
// for its output (independent loads; one thread walking 2048 dependent ones took 80 us), then the group sums are added in group
// order — a fixed order, so the result is bit-identical run to run
template <class T, int TM, int TN>
__global__ void __launch_bounds__(MD_BLOCK) k_longk_finish(LongKArgs g) {
  constexpr int P = TM * TN, G = MD_BLOCK / P;
  __shared__ T sums[G][P];
  const int tiles = g.tiles_m * g.tiles_n, bz = blockIdx.x / tiles, tile = blockIdx.x - bz * tiles;
  const int p = threadIdx.x % P, q = threadIdx.x / P;
  const T *part = (const T *)g.partial + (int64_t)blockIdx.x * g.nb * P;
  if (q < G) {
    T v = T(0);
    for (int b = q; b < g.nb; b += G) v = BAdd::apply(v, part[(int64_t)b * P + p]);
    sums[q][p] = v;
  }
  __syncthreads();
  if (threadIdx.x < P) {
    const int m = (tile / g.tiles_n) * TM + threadIdx.x / TN, n = (tile % g.tiles_n) * TN + threadIdx.x % TN;
    if (m < g.M && n < g.N) {
      T v = sums[0][threadIdx.x];
      for (int qq = 1; qq < G; ++qq) v = BAdd::apply(v, sums[qq][threadIdx.x]);
      ((T *)g.C)[(int64_t)bz * g.c_bs + (int64_t)m * g.c_ms + (int64_t)n * g.c_ns] = v;
    }
  }
}
template <class T, int TM, int TN> int longk_run(LongKArgs a, int64_t batch) {
  a.tiles_m = (a.M + TM - 1) / TM;
  a.tiles_n = (a.N + TN - 1) / TN;
  const int64_t patches = batch * a.tiles_m * a.tiles_n;
  if (patches > 65535) return -1;
  // blocks of >= 4096 k each (sixteen trips of the 256 threads), about 1024 blocks in all (four per CU)
  int64_t nb = (a.K + 4095) / 4096;
  const int64_t cap = patches >= 1024 ? 1 : 1024 / patches;
  if (nb > cap) nb = cap;
  a.chunk = (a.K + nb - 1) / nb;
  a.nb = (int)((a.K + a.chunk - 1) / a.chunk);
  void *partial = nullptr;
  MD_TRY(mdhip_alloc((size_t)patches * a.nb * TM * TN * sizeof(T), &partial));
  a.partial = partial;
  hipStream_t st = md_stream();
  MD_LAUNCH((k_longk_partial<T, TM, TN>), dim3((unsigned)a.nb, (unsigned)patches), MD_BLOCK, a);
  k_longk_finish<T, TM, TN><<<(unsigned)patches, MD_BLOCK, 0, st>>>(a);
  const int rc = MD_LAUNCH_CHECK("matmul(few outputs, long k)");
  mdhip_free(partial);
  return rc;
}
template <class T> int longk_dispatch(const LongKArgs &a, int64_t batch) {
  if (a.M == 1 && a.N == 1) return longk_run<T, 1, 1>(a, batch);
  if (a.M <= 2 && a.N <= 2) return longk_run<T, 2, 2>(a, batch);
  if (a.M <= 4 && a.N <= 4) return longk_run<T, 4, 4>(a, batch);
  if (a.N == 1) return longk_run<T, 8, 1>(a, batch);
  if (a.M == 1) return longk_run<T, 1, 8>(a, batch);
  return longk_run<T, 8, 8>(a, batch);
}
}  // namespace

// -1: not such a product; otherwise the launch status. dtype: MDHIP_F32 / F64 / I32 / I64.
int md_gemm_longk(const MdGemm &g, int dtype) {
  if (!md_opt(MD_OPT_GEMM_SKINNY) || g.M < 1 || g.N < 1 || g.M > 128 || g.N > 128 || g.batch < 1) return -1;
  const int64_t big = g.M > g.N ? g.M : g.N, small = g.M > g.N ? g.N : g.M;
  if (big <= 8) {
    if (g.K < 512) return -1;
  } else {
    // larger outputs: only while k dwarfs them (the tile kernels would leave most of the chip idle); float32 from 64 x 64 up has the
    // MFMA split-k path
    if (g.K < 8192 || g.K < 64 * big) return -1;
    if (dtype == MDHIP_F32 && small >= 64) return -1;
  }
  LongKArgs a{};
  a.A = g.a; a.B = g.b; a.C = g.c;
  a.K = g.K;
  a.M = (int)g.M; a.N = (int)g.N;
  a.a_bs = g.a_bs; a.a_ms = g.a_ms; a.a_ks = g.a_ks;
  a.b_bs = g.b_bs; a.b_ks = g.b_ks; a.b_ns = g.b_ns;
  a.c_bs = g.c_bs; a.c_ms = g.c_ms; a.c_ns = g.c_ns;
  switch (dtype) {
    case MDHIP_F32: return longk_dispatch<float>(a, g.batch);
    case MDHIP_F64: return longk_dispatch<double>(a, g.batch);
    case MDHIP_I32: return longk_dispatch<int32_t>(a, g.batch);
    case MDHIP_I64: return longk_dispatch<int64_t>(a, g.batch);
  }
  return -1;
}

// -1: not a product for these kernels (the caller goes on to the MFMA / generic paths); otherwise the launch status.
// dtype: MDHIP_F32 / MDHIP_F64.
int md_gemm_skinny(const MdGemm &g, int dtype) {
  if (!md_opt(MD_OPT_GEMM_SKINNY) || (dtype != MDHIP_F32 && dtype != MDHIP_F64) || g.batch < 1 || g.batch > 65535) return -1;
  const int64_t esz = dtype == MDHIP_F32 ? 4 : 8, V = 16 / esz;
  SkinnyArgs a{};
  if (g.N <= 8 && g.M > 8) {          // thin N: X = A, Y = B, out = C
    a.X = g.a; a.Y = g.b; a.out = g.c;
    a.R = g.M; a.K = g.K; a.nc = (int)g.N;
    a.xr = g.a_ms; a.xk = g.a_ks; a.x_bs = g.a_bs;
    a.yk = g.b_ks; a.yc = g.b_ns; a.y_bs = g.b_bs;
    a.o_r = g.c_ms; a.o_c = g.c_ns; a.o_bs = g.c_bs;
  } else if (g.M <= 8 && g.N > 8) {   // thin M: the transposed problem, X = B^T, Y = A^T, out = C^T
    a.X = g.b; a.Y = g.a; a.out = g.c;
    a.R = g.N; a.K = g.K; a.nc = (int)g.M;
    a.xr = g.b_ns; a.xk = g.b_ks; a.x_bs = g.b_bs;
    a.yk = g.a_ks; a.yc = g.a_ms; a.y_bs = g.a_bs;
    a.o_r = g.c_ns; a.o_c = g.c_ms; a.o_bs = g.c_bs;
  } else {
    return -1;
  }
  // worth a streaming kernel: a big operand of at least 1 Mi elements per launch, long enough both ways to fill the chip
  if (a.R < 512 || a.K < 64 || a.R * a.K * g.batch < (1ll << 20)) return -1;
  if (((uintptr_t)a.X & 15) || (a.x_bs % V)) return -1;
  bool rowdot;
  if (a.xk == 1 && a.xr >= a.K && a.xr % V == 0 && a.K % V == 0) rowdot = true;
  else if (a.xr == 1 && a.xk >= a.R && a.xk % V == 0 && a.R % V == 0) rowdot = false;
  else return -1;
  return dtype == MDHIP_F32 ? skinny_run<float>(a, g.batch, rowdot) : skinny_run<double>(a, g.batch, rowdot);
}
