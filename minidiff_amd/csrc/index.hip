// index.hip — gather / scatter over an index plan (bit-exact data movement).
//
// Serves reference minidiff/backend/numpy.py:73-75 (a[key] with integer-array
// keys), :105 (np.add.at, the backward of getitem: definitions.py:186-189),
// :108 / :124 (take_along_axis / put_along_axis, the max/min gradients
// definitions.py:98-127) and tensor.py:376-379 (__setitem__, used by the
// finite-difference checker utils.py:147-148).
//
// Payload bytes are moved, never recomputed, so results are bit-identical to
// NumPy's. np.add.at accumulates duplicates in index order; float addition is
// not associative, and a[key] = v keeps the LAST duplicate, so both walk positions
// sequentially when the plan is small and otherwise run order-preserving rounds
// (per destination the smallest unfinished position goes first) — no float
// atomics, no write races. Integer ADD uses atomics (exact in any order).
#include <cstring>


#include "md_hip.h"
#include "md_rng.h"

extern "C" int mdhip_alloc(size_t, void **);
extern "C" int mdhip_free(void *);

namespace {

template <class E>
__global__ void __launch_bounds__(MD_BLOCK) k_gather(mdhip_index_plan pl, int64_t total, const E *src, E *out, MdIter oit, int *err) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    const int64_t off = md_plan_offset(pl, i, pos, &oob);
    if (oob) { *err = 1; continue; }
    int64_t oo = 0;
    for (int d = 0; d < pl.ndim; ++d) oo += pos[d] * oit.strides[0][d];
    out[oo] = src[off];
  }
}

// Contiguous runs (a[idx] selecting whole rows, take along an outer axis): when the last plan
// axis is contiguous on both sides and no index array varies along it, 16-B units move instead
// of elements; `vshift` = log2(elements per unit).
__global__ void __launch_bounds__(MD_BLOCK) k_gather_vec(mdhip_index_plan pl, int64_t total_v, int vshift, const uint4 *__restrict__ src,
                                                        uint4 *__restrict__ out, MdIter oit, int *err) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_v; i += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    const int64_t off = md_plan_offset(pl, i << vshift, pos, &oob);
    if (oob) { *err = 1; continue; }
    int64_t oo = 0;
    for (int d = 0; d < pl.ndim; ++d) oo += pos[d] * oit.strides[0][d];
    out[oo >> vshift] = src[off >> vshift];
  }
}

// LONG runs (rows of >= 64 units: an embedding lookup W[idx], a[idx] selecting whole rows of a wide matrix): the plan arithmetic of
// a run — the div / mod chain over the plan's axes and the index loads — is done ONCE per run by the wave that copies it, and the
// run itself moves as coalesced 16-B units, four in flight per lane. k_gather_vec above pays that arithmetic per 16-B unit
// (8192 rows x 16 KiB: 79 us; a plain copy of the same bytes takes 48).
__global__ void __launch_bounds__(MD_BLOCK) k_gather_runs(mdhip_index_plan pl, int64_t n_runs, int64_t run_elems, int vshift, const uint4 *__restrict__ src,
                                                         uint4 *__restrict__ out, MdIter oit, int *err) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t lu = run_elems >> vshift;   // units per run
  for (int64_t p = wave; p < n_runs; p += n_waves) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    const int64_t off = md_plan_offset(pl, p * run_elems, pos, &oob);   // (the same for every lane of the wave)
    if (oob) {
      if (lane == 0) *err = 1;
      continue;
    }
    int64_t oo = 0;
    for (int d = 0; d < pl.ndim; ++d) oo += pos[d] * oit.strides[0][d];
    const uint4 *s = src + (off >> vshift);
    uint4 *o = out + (oo >> vshift);
    int64_t u = lane;
    for (; u + 192 < lu; u += 256) {
      const uint4 a = s[u], b = s[u + 64], c = s[u + 128], d = s[u + 192];
      o[u] = a; o[u + 64] = b; o[u + 128] = c; o[u + 192] = d;
    }
    for (; u < lu; u += 64) o[u] = s[u];
  }
}

// `step` > 1: only every step-th position is probed (run plans: no index varies along the last axis)
__global__ void __launch_bounds__(MD_BLOCK) k_check_bounds(mdhip_index_plan pl, int64_t count, int64_t step, int *err) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    md_plan_offset(pl, i * step, pos, &oob);
    if (oob) *err = 1;
  }
}

// Run plans (whole rows): the bounds probe and a census of the destination rows in one pass — a row hit twice sets bit 1 of the
// flag. Unique rows (a permutation, distinct token ids) need no ordering at all: the sort of scatter_runs is skipped for them.
__global__ void __launch_bounds__(MD_BLOCK) k_check_bounds_dups(mdhip_index_plan pl, int64_t P, int64_t L, int64_t lo, int64_t unit, int *cnt, int *flag) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    const int64_t off = md_plan_offset(pl, p * L, pos, &oob);
    if (oob) { atomicOr(flag, 1); continue; }
    if (atomicAdd(cnt + (off - lo) / unit, 1) > 0) atomicOr(flag, 2);
  }
}

struct ValDesc {
  const void *p;
  int is_scalar;
  int64_t strides[MDHIP_MAX_NDIM];
  // CAPTURED scatters (no host read of the bounds verdict): the flag of the bounds pass; bit 0 set = some index is out of range,
  // and every kernel that would write leaves at once — NumPy's "raises before touching the destination", minus the raise, which
  // waits for the next synchronisation (md_sticky_check). nullptr outside a capture: the host has already looked.
  const int *guard;
};
#define MD_SCATTER_GUARD(v) do { if ((v).guard && (*(v).guard & 1)) return; } while (0)
__global__ void k_flag_to_sticky(const int *flag, int *sticky) { if (*flag & 1) *sticky = 1; }
template <class T> __device__ __forceinline__ T val_at(const ValDesc &v, T s, const mdhip_index_plan &pl, const int64_t *pos) {
  if (v.is_scalar) return s;
  int64_t vo = 0;
  for (int d = 0; d < pl.ndim; ++d) vo += pos[d] * v.strides[d];
  return ((const T *)v.p)[vo];
}

// integer ADD: atomics are exact and order-independent
template <class T>
__global__ void __launch_bounds__(MD_BLOCK) k_scatter_add_int(mdhip_index_plan pl, int64_t total, T *dst, ValDesc v, T s) {
  MD_SCATTER_GUARD(v);
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    const int64_t off = md_plan_offset(pl, i, pos, &oob);
    const T val = val_at<T>(v, s, pl, pos);
    if constexpr (sizeof(T) == 8) atomicAdd((unsigned long long *)(dst + off), (unsigned long long)val);
    else atomicAdd((unsigned int *)(dst + off), (unsigned int)val);
  }
}
// one lane, positions in order: exactly np.add.at / a[key] = v
template <class T, int MODE>
__global__ void k_scatter_serial(mdhip_index_plan pl, int64_t total, T *dst, ValDesc v, T s) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  MD_SCATTER_GUARD(v);
  for (int64_t i = 0; i < total; ++i) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    const int64_t off = md_plan_offset(pl, i, pos, &oob);
    const T val = val_at<T>(v, s, pl, pos);
    if constexpr (MODE == MDHIP_SCATTER_ADD) {
      if constexpr (md_same<T, uint8_t>::value) dst[off] = (uint8_t)(dst[off] || val);
      else dst[off] = md_storage_add(dst[off], val);
    } else {
      dst[off] = val;
    }
  }
}
// float ADD at scale, order-preserving without atomics on the payload:
// round r: every unfinished position p bids atomicMin(owner[slot(off)], p); the
// winner for its destination is the smallest unfinished p, applies its add, and
// retires. Rounds = max multiplicity of a destination (1 when keys are unique).
__global__ void __launch_bounds__(MD_BLOCK) k_offsets(mdhip_index_plan pl, int64_t total, int64_t *offs) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    offs[i] = md_plan_offset(pl, i, pos, &oob);
  }
}
__global__ void __launch_bounds__(MD_BLOCK) k_bid(const int64_t *offs, const uint8_t *done, int64_t total, unsigned long long *owner, int64_t nslots) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    if (done[i]) continue;
    const uint64_t slot = ((uint64_t)offs[i] * 0x9E3779B97F4A7C15ull) % (uint64_t)nslots;
    atomicMin(owner + slot, (unsigned long long)i);
  }
}
template <class T, int MODE>
__global__ void __launch_bounds__(MD_BLOCK) k_apply(mdhip_index_plan pl, const int64_t *offs, uint8_t *done, int64_t total,
                                                   const unsigned long long *owner, int64_t nslots, T *dst, ValDesc v, T s, int *remaining) {
  MD_SCATTER_GUARD(v);   // (deferred bounds verdict, option index_defer: nothing is written once the bounds pass found a bad index)
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    if (done[i]) continue;
    const uint64_t slot = ((uint64_t)offs[i] * 0x9E3779B97F4A7C15ull) % (uint64_t)nslots;
    if (owner[slot] == (unsigned long long)i) {
      int64_t pos[MDHIP_MAX_NDIM];
      int64_t lin = i;
      for (int d = pl.ndim - 1; d >= 0; --d) { int64_t e = pl.shape[d]; int64_t q = lin / e; pos[d] = lin - q * e; lin = q; }
      const T val = val_at<T>(v, s, pl, pos);
      if constexpr (MODE == MDHIP_SCATTER_ADD) dst[offs[i]] = md_storage_add(dst[offs[i]], val);
      else dst[offs[i]] = val;
      done[i] = 1;
    } else {
      *remaining = 1;
    }
  }
}

static int read_flag(int *dflag, int *host) {
  MD_TRY(md_hip_check(hipMemcpyAsync(host, dflag, sizeof(int), hipMemcpyDeviceToHost, md_stream()), "hipMemcpyAsync(flag)"));
  return md_hip_check(hipStreamSynchronize(md_stream()), "hipStreamSynchronize");
}

// ---- contiguous runs: order at ROW granularity -------------------------------------------
// The backward of a[idx] (np.add.at(grad, idx, g), definitions.py:186-189) and a[idx] = v move
// whole rows: the plan's last axis is a contiguous run of the destination that no index array
// varies along. Then the duplicates question is per ROW: destinations of two plan rows are
// either identical or disjoint (checked on the host: every outer stride is a multiple of a
// common g >= run length). The P row offsets are sorted stably (the LSD radix sort below, so equal
// destinations keep plan order) and each destination row is produced by ONE pass that applies
// its contributions in that order: np.add.at's accumulation order / last-write-wins for SET,
// O(P log P + P*L) whatever the multiplicities.
__global__ void __launch_bounds__(MD_BLOCK) k_run_offsets(mdhip_index_plan pl, int64_t P, int64_t L, int64_t lo, int64_t unit, uint64_t *keys, int64_t *ids) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    // key = offset relative to the smallest reachable one, in units of the rows' common stride (every row offset is a multiple of
    // it: run_geometry) — 8192 rows of 4096 elements sort on 13 bits, two radix passes, instead of 25 bits, four
    keys[p] = (uint64_t)((md_plan_offset(pl, p * L, pos, &oob) - lo) / unit);
    ids[p] = p;
  }
}
template <class T, int MODE>
__global__ void __launch_bounds__(MD_BLOCK) k_run_apply(mdhip_index_plan pl, int64_t P, int64_t L, int64_t lo, int64_t unit, const uint64_t *__restrict__ keys,
                                                       const int64_t *__restrict__ ids, T *dst, ValDesc v, T s) {
  MD_SCATTER_GUARD(v);
  const int64_t gs = (int64_t)gridDim.x * blockDim.x, total = P * L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    const int64_t q = i / L, c = i - q * L;
    const uint64_t key = keys[q];
    if (q > 0 && keys[q - 1] == key) continue;  // not the first contribution of its destination row
    T *d = dst + ((int64_t)key * unit + lo + c);
    auto value = [&](int64_t p) -> T {
      if (v.is_scalar) return s;
      int64_t lin = p, vo = c * v.strides[pl.ndim - 1];
      for (int dd = pl.ndim - 2; dd >= 0; --dd) {
        const int64_t e = pl.shape[dd], qq = lin / e;
        vo += (lin - qq * e) * v.strides[dd];
        lin = qq;
      }
      return ((const T *)v.p)[vo];
    };
    if constexpr (MODE == MDHIP_SCATTER_ADD) {
      T acc = *d;
      for (int64_t q2 = q; q2 < P && keys[q2] == key; ++q2) acc = md_storage_add(acc, value(ids[q2]));
      *d = acc;
    } else {
      int64_t q2 = q;
      while (q2 + 1 < P && keys[q2 + 1] == key) ++q2;
      *d = value(ids[q2]);
    }
  }
}
// the same with 16-B units when rows, destination and values are 16-B aligned
template <class T, int MODE>
__global__ void __launch_bounds__(MD_BLOCK) k_run_apply_vec(mdhip_index_plan pl, int64_t P, int64_t L, int64_t lo, int64_t unit, const uint64_t *__restrict__ keys,
                                                           const int64_t *__restrict__ ids, T *dst, ValDesc v, T s) {
  constexpr int V = 16 / sizeof(T);
  MD_SCATTER_GUARD(v);
  const int64_t Lv = L / V, gs = (int64_t)gridDim.x * blockDim.x, total = P * Lv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    const int64_t q = i / Lv, c = (i - q * Lv) * V;
    const uint64_t key = keys[q];
    if (q > 0 && keys[q - 1] == key) continue;
    MdVec<T, V> *d = reinterpret_cast<MdVec<T, V> *>(dst + ((int64_t)key * unit + lo + c));
    auto value = [&](int64_t p) -> MdVec<T, V> {
      MdVec<T, V> r;
      if (v.is_scalar) {
#pragma unroll
        for (int j = 0; j < V; ++j) r.v[j] = s;
        return r;
      }
      int64_t lin = p, vo = c;
      for (int dd = pl.ndim - 2; dd >= 0; --dd) {
        const int64_t e = pl.shape[dd], qq = lin / e;
        vo += (lin - qq * e) * v.strides[dd];
        lin = qq;
      }
      return *reinterpret_cast<const MdVec<T, V> *>((const T *)v.p + vo);
    };
    if constexpr (MODE == MDHIP_SCATTER_ADD) {
      // contributions are ADDED in plan order (np.add.at), but their loads need not wait for one another: four in flight
      // (128 contributions per destination row: 198 us with one dependent load at a time)
      MdVec<T, V> acc = *d;
      int64_t n = 1;
      while (q + n < P && keys[q + n] == key) ++n;
      int64_t j0 = 0;
      for (; j0 + 4 <= n; j0 += 4) {
        const MdVec<T, V> t0 = value(ids[q + j0]), t1 = value(ids[q + j0 + 1]), t2 = value(ids[q + j0 + 2]), t3 = value(ids[q + j0 + 3]);
#pragma unroll
        for (int j = 0; j < V; ++j) acc.v[j] = BAdd::apply(BAdd::apply(BAdd::apply(BAdd::apply(acc.v[j], t0.v[j]), t1.v[j]), t2.v[j]), t3.v[j]);
      }
      for (; j0 < n; ++j0) {
        const MdVec<T, V> t = value(ids[q + j0]);
#pragma unroll
        for (int j = 0; j < V; ++j) acc.v[j] = BAdd::apply(acc.v[j], t.v[j]);
      }
      *d = acc;
    } else {
      int64_t q2 = q;
      while (q2 + 1 < P && keys[q2 + 1] == key) ++q2;
      *d = value(ids[q2]);
    }
  }
}
template <class T> static bool run_vectorisable(const mdhip_index_plan *pl, int64_t L, const void *dst, const ValDesc &v) {
  constexpr int V = 16 / sizeof(T);
  if (V < 2 || (L % V) || ((uintptr_t)dst & 15)) return false;
  const int nd = pl->ndim;
  for (int d = 0; d < nd - 1; ++d)
    if (pl->shape[d] > 1 && (pl->src_strides[d] % V)) return false;
  for (int k = 0; k < pl->n_idx; ++k)
    if (pl->idx_extent[k] > 1 && (pl->idx_mult[k] % V)) return false;
  if (!v.is_scalar) {
    if (v.strides[nd - 1] != 1 || ((uintptr_t)v.p & 15)) return false;
    for (int d = 0; d < nd - 1; ++d)
      if (pl->shape[d] > 1 && (v.strides[d] % V)) return false;
  }
  return true;
}
static int64_t gcd64(int64_t a, int64_t b) {
  if (a < 0) a = -a;
  if (b < 0) b = -b;
  while (b) { const int64_t t = a % b; a = b; b = t; }
  return a;
}
static bool run_geometry(const mdhip_index_plan *pl, int64_t *L, int64_t *P, int64_t *unit = nullptr) {
  const int nd = pl->ndim;
  if (nd < 1) return false;
  *L = pl->shape[nd - 1];
  if (*L < 8 || pl->src_strides[nd - 1] != 1) return false;
  for (int k = 0; k < pl->n_idx; ++k)
    if (pl->idx_strides[k][nd - 1] != 0) return false;
  int64_t g = 0;
  *P = 1;
  for (int d = 0; d < nd - 1; ++d) {
    *P *= pl->shape[d];
    if (pl->shape[d] > 1) g = gcd64(g, pl->src_strides[d]);
  }
  for (int k = 0; k < pl->n_idx; ++k)
    if (pl->idx_extent[k] > 1) g = gcd64(g, pl->idx_mult[k]);
  if (unit) *unit = g > 0 ? g : 1;
  return (g == 0 || g >= *L) && *P < (1ll << 27);  // (the sort keeps 256 counters per 2048 rows: 64 MiB at this bound)
}
// ---- stable LSD radix sort of (key, id) pairs: 8 bits per pass, as many passes as the keys have bits --------------
// Pass = histogram per 2048-item tile -> exclusive scan of the [digit][tile] counts -> scatter. Stability inside a
// tile: items are taken in 8 rounds of 256 (round-major = input order), the four waves of a round one after the other;
// inside a wave a lane's rank among the lanes with the same digit comes from 8 ballots (one per digit bit).
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = MD_BLOCK * RS_ITEMS;

__global__ void __launch_bounds__(MD_BLOCK) k_rs_hist(const uint64_t *__restrict__ keys, int64_t P, int shift, uint32_t *__restrict__ hist, int64_t nblk) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const int64_t i = base + r * MD_BLOCK + threadIdx.x;
    if (i < P) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of n counters, three launches: per-chunk scan (2048 per block) + chunk totals, scan of the totals (one
// block), add-back
__global__ void __launch_bounds__(MD_BLOCK) k_rs_scan_chunks(uint32_t *__restrict__ a, int64_t n, uint32_t *__restrict__ totals) {
  __shared__ uint32_t part[MD_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)threadIdx.x * RS_ITEMS;
  uint32_t v[RS_ITEMS], sum = 0;
#pragma unroll
  for (int j = 0; j < RS_ITEMS; ++j) {
    v[j] = base + j < n ? a[base + j] : 0u;
    sum += v[j];
  }
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int d = 1; d < MD_BLOCK; d <<= 1) {  // inclusive Hillis-Steele over the 256 thread sums
    const uint32_t t = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
#pragma unroll
  for (int j = 0; j < RS_ITEMS; ++j) {
    if (base + j < n) a[base + j] = run;
    run += v[j];
  }
  if (threadIdx.x == MD_BLOCK - 1) totals[blockIdx.x] = part[MD_BLOCK - 1];
}
__global__ void __launch_bounds__(MD_BLOCK) k_rs_scan_totals(uint32_t *__restrict__ totals, int64_t m) {
  __shared__ uint32_t part[MD_BLOCK];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t c0 = 0; c0 < m; c0 += MD_BLOCK) {
    const int64_t i = c0 + threadIdx.x;
    const uint32_t v = i < m ? totals[i] : 0u;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < MD_BLOCK; d <<= 1) {
      const uint32_t t = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < m) totals[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += part[MD_BLOCK - 1];
    __syncthreads();
  }
}
__global__ void __launch_bounds__(MD_BLOCK) k_rs_scan_add(uint32_t *__restrict__ a, int64_t n, const uint32_t *__restrict__ totals) {
  const uint32_t off = totals[blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * RS_TILE + (int64_t)threadIdx.x * RS_ITEMS;
#pragma unroll
  for (int j = 0; j < RS_ITEMS; ++j)
    if (base + j < n) a[base + j] += off;
}
__global__ void __launch_bounds__(MD_BLOCK) k_rs_scatter(const uint64_t *__restrict__ kin, const int64_t *__restrict__ iin, uint64_t *__restrict__ kout,
                                                        int64_t *__restrict__ iout, int64_t P, int shift, const uint32_t *__restrict__ offs, int64_t nblk) {
  __shared__ uint32_t cnt[256];  // next output slot of every digit for this tile
  cnt[threadIdx.x] = offs[(int64_t)threadIdx.x * nblk + blockIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  for (int r = 0; r < RS_ITEMS; ++r) {
    const int64_t i = base + r * MD_BLOCK + threadIdx.x;
    const bool valid = i < P;
    const uint64_t key = valid ? kin[i] : 0ull;
    const int64_t id = valid ? iin[i] : 0;
    const uint32_t digit = (uint32_t)(key >> shift) & 255u;
    uint32_t pos = 0;
    for (int w = 0; w < MD_BLOCK / 64; ++w) {
      if (wave == w) {
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const uint64_t vote = __ballot((digit >> b) & 1u);
          peers &= ((digit >> b) & 1u) ? vote : ~vote;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & below);
        if (valid) {
          pos = cnt[digit] + rank;                                        // every lane of the wave reads ...
          if (rank == 0) cnt[digit] += (uint32_t)__popcll(peers);          // ... before the group's first lane advances the slot
        }
      }
      __syncthreads();
    }
    if (valid) {
      kout[pos] = key;
      iout[pos] = id;
    }
  }
}
// keys/ids: two halves of 2P elements each (ping-pong); returns which half holds the sorted sequence
static int radix_sort_pairs(uint64_t *keys, int64_t *ids, int64_t P, int key_bits, int *result_half) {
  hipStream_t st = md_stream();
  const int64_t nblk = (P + RS_TILE - 1) / RS_TILE;
  const int64_t n = 256 * nblk, nchunks = (n + RS_TILE - 1) / RS_TILE;
  void *hist = nullptr, *totals = nullptr;
  MD_TRY(mdhip_alloc((size_t)n * 4, &hist));
  int rc = mdhip_alloc((size_t)nchunks * 4, &totals);
  int half = 0;
  if (rc == MDHIP_OK) {
    for (int shift = 0; shift < key_bits; shift += 8) {
      uint64_t *kin = keys + half * P, *kout = keys + (half ^ 1) * P;
      int64_t *iin = ids + half * P, *iout = ids + (half ^ 1) * P;
      k_rs_hist<<<(unsigned)nblk, MD_BLOCK, 0, st>>>(kin, P, shift, (uint32_t *)hist, nblk);
      k_rs_scan_chunks<<<(unsigned)nchunks, MD_BLOCK, 0, st>>>((uint32_t *)hist, n, (uint32_t *)totals);
      k_rs_scan_totals<<<1, MD_BLOCK, 0, st>>>((uint32_t *)totals, nchunks);
      k_rs_scan_add<<<(unsigned)nchunks, MD_BLOCK, 0, st>>>((uint32_t *)hist, n, (const uint32_t *)totals);
      k_rs_scatter<<<(unsigned)nblk, MD_BLOCK, 0, st>>>(kin, iin, kout, iout, P, shift, (const uint32_t *)hist, nblk);
      half ^= 1;
    }
    rc = MD_LAUNCH_CHECK("scatter(runs): radix sort");
  }
  if (totals) mdhip_free(totals);
  mdhip_free(hist);
  *result_half = half;
  return rc;
}

// key range of a run plan (host side): row offsets lie in [lo, hi]
static void run_key_range(const mdhip_index_plan *pl, int64_t L, int64_t *lo_out, int64_t *hi_out) {
  int64_t lo = 0, hi = L - 1;
  for (int d = 0; d < pl->ndim - 1; ++d) {
    const int64_t e = (pl->shape[d] - 1) * pl->src_strides[d];
    if (e < 0) lo += e; else hi += e;
  }
  for (int k = 0; k < pl->n_idx; ++k) {
    const int64_t e = (pl->idx_extent[k] - 1) * pl->idx_mult[k];
    if (e < 0) lo += e; else hi += e;
  }
  *lo_out = lo;
  *hi_out = hi;
}

// `unique`: the census of the bounds pass found no destination row twice — plan order is irrelevant, no sort
template <class T, int MODE, bool VEC_OK = true>
static int scatter_runs(const mdhip_index_plan *pl, int64_t P, int64_t L, void *dst, const ValDesc &v, T s, bool unique) {
  hipStream_t st = md_stream();
  // keys are offset - lo, so the sort needs bits(hi - lo) only
  int64_t lo, hi;
  run_key_range(pl, L, &lo, &hi);
  int64_t L2 = 0, P2 = 0, unit = 1;
  run_geometry(pl, &L2, &P2, &unit);
  int key_bits = 1;
  while (key_bits < 64 && ((uint64_t)((hi - lo) / unit) >> key_bits) != 0) ++key_bits;
  void *keys = nullptr, *ids = nullptr;
  MD_TRY(mdhip_alloc((size_t)P * 16, &keys));
  int rc = mdhip_alloc((size_t)P * 16, &ids);
  if (rc == MDHIP_OK) {
    uint64_t *kin = (uint64_t *)keys;
    int64_t *iin = (int64_t *)ids;
    k_run_offsets<<<md_grid_for(P), MD_BLOCK, 0, st>>>(*pl, P, L, lo, unit, kin, iin);
    int half = 0;
    if (!unique) rc = radix_sort_pairs(kin, iin, P, key_bits, &half);   // (unique rows: the apply kernels only ever look at EQUAL neighbouring keys)
    if (rc == MDHIP_OK) {
      const uint64_t *kout = kin + half * P;
      const int64_t *iout = iin + half * P;
      // one item per thread, no grid-stride trips: only the FIRST plan row of a destination does that destination's work, so with
      // duplicates the busy items are few and clustered — under a capped grid a trip held 4 of 64 busy rows (16 blocks at work,
      // 16 trips one after the other: 0.9 TB/s); every other block exits at once
      auto full_grid = [](int64_t items) { const int64_t b = (items + MD_BLOCK - 1) / MD_BLOCK; return (unsigned)(b < 1 ? 1 : (b > 0x7fffffffll ? 0x7fffffffll : b)); };
      bool vec = false;
      if constexpr (VEC_OK) vec = run_vectorisable<T>(pl, L, dst, v);
      if constexpr (VEC_OK) {
        if (vec) k_run_apply_vec<T, MODE><<<full_grid(P * (L / (16 / (int64_t)sizeof(T)))), MD_BLOCK, 0, st>>>(*pl, P, L, lo, unit, kout, iout, (T *)dst, v, s);
      }
      if (!vec) k_run_apply<T, MODE><<<full_grid(P * L), MD_BLOCK, 0, st>>>(*pl, P, L, lo, unit, kout, iout, (T *)dst, v, s);
      rc = MD_LAUNCH_CHECK("scatter(runs)");
    }
  }
  if (ids) mdhip_free(ids);
  mdhip_free(keys);
  return rc;
}

// ---- element-granular order: sort the positions by destination, one serial pass per destination -----------------------
// np.add.at with duplicate destinations at ELEMENT granularity (a histogram: a million contributions into ten bins) and a[idx] = v
// with repeats. The positions' destination offsets are sorted stably (plan order survives among equal destinations), then the
// thread that holds the FIRST position of a destination applies that destination's contributions one after the other — NumPy's
// order, O(n log n) whatever the multiplicities, no host round trip (capturable). The bid / apply rounds below (one round, one
// host read-back, per multiplicity level: a million rounds for a million contributions to one element) remain only for plans
// beyond the sort's 2^27 positions.
__global__ void __launch_bounds__(MD_BLOCK) k_elem_offsets(mdhip_index_plan pl, int64_t total, int64_t lo, uint64_t *keys, int64_t *ids) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    int64_t pos[MDHIP_MAX_NDIM];
    bool oob = false;
    keys[i] = (uint64_t)(md_plan_offset(pl, i, pos, &oob) - lo);
    ids[i] = i;
  }
}
template <class T> __device__ __forceinline__ T elem_value(const mdhip_index_plan &pl, const ValDesc &v, T s, int64_t p) {
  if (v.is_scalar) return s;
  int64_t lin = p, vo = 0;
  for (int dd = pl.ndim - 1; dd >= 0; --dd) {
    const int64_t e = pl.shape[dd], qq = lin / e;
    vo += (lin - qq * e) * v.strides[dd];
    lin = qq;
  }
  return ((const T *)v.p)[vo];
}
template <class T> __device__ __forceinline__ T elem_add(T acc, T x) {
  if constexpr (md_same<T, uint8_t>::value) return (uint8_t)(acc || x);   // (bool: np.add on booleans is logical or)
  else return md_storage_add(acc, x);
}
constexpr int ELEM_SHORT = 64;   // destinations with more contributions than this go to the wave-per-destination kernel
// One thread per sorted position. SET: the LAST position of a destination writes (no scan). ADD: the FIRST position of a destination
// adds its contributions in order when they are few; a long destination (a histogram bin) is put on a work list instead — one lane
// walking 10^6 dependent loads took 0.4 s.
template <class T, int MODE>
__global__ void __launch_bounds__(MD_BLOCK) k_elem_apply(mdhip_index_plan pl, int64_t total, int64_t lo, const uint64_t *__restrict__ keys,
                                                        const int64_t *__restrict__ ids, T *dst, ValDesc v, T s, int64_t *long_list, int *n_long) {
  MD_SCATTER_GUARD(v);
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= total) return;
  const uint64_t key = keys[q];
  T *d = dst + ((int64_t)key + lo);
  if constexpr (MODE == MDHIP_SCATTER_SET) {
    if (q + 1 == total || keys[q + 1] != key) *d = elem_value<T>(pl, v, s, ids[q]);
  } else {
    if (q > 0 && keys[q - 1] == key) return;   // not the first contribution of its destination
    int64_t n = 1;
    while (n <= ELEM_SHORT && q + n < total && keys[q + n] == key) ++n;
    if (n > ELEM_SHORT) {
      long_list[atomicAdd(n_long, 1)] = q;
      return;
    }
    T acc = *d;
    for (int64_t j = 0; j < n; ++j) acc = elem_add<T>(acc, elem_value<T>(pl, v, s, ids[q + j]));
    *d = acc;
  }
}
// A wave per long destination: 64 contributions are LOADED at once (their addresses do not depend on one another), staged in LDS, and
// added by lane 0 in plan order — the order of np.add.at, at ~10 ns instead of ~400 ns per contribution.
template <class T>
__global__ void __launch_bounds__(MD_BLOCK) k_elem_apply_long(mdhip_index_plan pl, int64_t total, int64_t lo, const uint64_t *__restrict__ keys,
                                                             const int64_t *__restrict__ ids, T *dst, ValDesc v, T s, const int64_t *long_list,
                                                             const int *n_long) {
  MD_SCATTER_GUARD(v);
  __shared__ T stage[MD_BLOCK / 64][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t count = *n_long;
  for (int64_t e = wave; e < count; e += n_waves) {
    const int64_t q = long_list[e];
    const uint64_t key = keys[q];
    T *d = dst + ((int64_t)key + lo);
    T acc = *d;
    for (int64_t base = q;; base += 64) {
      const int64_t p = base + lane;
      const bool mine = p < total && keys[p] == key;
      if (mine) stage[w][lane] = elem_value<T>(pl, v, s, ids[p]);
      const unsigned long long live = __ballot(mine);
      const int n = live == ~0ull ? 64 : __builtin_ctzll(~live);   // (sorted: the destination's positions are a prefix of the chunk)
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) {
        if (n == 64) {   // whole chunk: the 64 LDS reads issue together, only the adds are a chain
#pragma unroll
          for (int j = 0; j < 64; ++j) acc = elem_add<T>(acc, stage[w][j]);
        } else {
          for (int j = 0; j < n; ++j) acc = elem_add<T>(acc, stage[w][j]);
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (n < 64) break;
    }
    if (lane == 0) *d = acc;
  }
}
// every destination offset of the plan lies in [lo, hi] (host side, from the extents)
static void plan_offset_range(const mdhip_index_plan *pl, int64_t *lo_out, int64_t *hi_out) {
  int64_t lo = 0, hi = 0;
  for (int d = 0; d < pl->ndim; ++d) {
    const int64_t e = (pl->shape[d] - 1) * pl->src_strides[d];
    if (e < 0) lo += e; else hi += e;
  }
  for (int k = 0; k < pl->n_idx; ++k) {
    const int64_t e = (pl->idx_extent[k] - 1) * pl->idx_mult[k];
    if (e < 0) lo += e; else hi += e;
  }
  *lo_out = lo;
  *hi_out = hi;
}
template <class T, int MODE>
static int scatter_ordered(const mdhip_index_plan *pl, int64_t total, void *dst, const ValDesc &v, T s);
template <class T, int MODE>
static int scatter_sorted(const mdhip_index_plan *pl, int64_t total, void *dst, const ValDesc &v, T s) {
  if (total >= (1ll << 27) || md_opt(MD_OPT_SCATTER_SORTED) == 0)   // (the sort keeps 256 counters per 2048 positions; option: A/B)
    return scatter_ordered<T, MODE>(pl, total, dst, v, s);
  hipStream_t st = md_stream();
  int64_t lo, hi;
  plan_offset_range(pl, &lo, &hi);
  int key_bits = 1;
  while (key_bits < 64 && ((uint64_t)(hi - lo) >> key_bits) != 0) ++key_bits;
  void *keys = nullptr, *ids = nullptr;
  MD_TRY(mdhip_alloc((size_t)total * 16, &keys));
  int rc = mdhip_alloc((size_t)total * 16, &ids);
  if (rc == MDHIP_OK) {
    k_elem_offsets<<<md_grid_for(total), MD_BLOCK, 0, st>>>(*pl, total, lo, (uint64_t *)keys, (int64_t *)ids);
    int half = 0;
    rc = radix_sort_pairs((uint64_t *)keys, (int64_t *)ids, total, key_bits, &half);
    void *long_list = nullptr, *n_long = nullptr;
    if (rc == MDHIP_OK && MODE == MDHIP_SCATTER_ADD) {   // at most total / (ELEM_SHORT + 1) long destinations
      rc = mdhip_alloc((size_t)(total / (ELEM_SHORT + 1) + 1) * 8, &long_list);
      if (rc == MDHIP_OK) rc = mdhip_alloc(sizeof(int), &n_long);
      if (rc == MDHIP_OK) (void)hipMemsetAsync(n_long, 0, sizeof(int), st);
    }
    if (rc == MDHIP_OK) {
      const int64_t blocks = (total + MD_BLOCK - 1) / MD_BLOCK;   // (< 2^19: one position per thread)
      const uint64_t *kout = (const uint64_t *)keys + half * total;
      const int64_t *iout = (const int64_t *)ids + half * total;
      k_elem_apply<T, MODE><<<(unsigned)blocks, MD_BLOCK, 0, st>>>(*pl, total, lo, kout, iout, (T *)dst, v, s, (int64_t *)long_list, (int *)n_long);
      if constexpr (MODE == MDHIP_SCATTER_ADD)
        k_elem_apply_long<T><<<1024, MD_BLOCK, 0, st>>>(*pl, total, lo, kout, iout, (T *)dst, v, s, (const int64_t *)long_list, (const int *)n_long);
      rc = MD_LAUNCH_CHECK("scatter(sorted)");
    }
    if (n_long) mdhip_free(n_long);
    if (long_list) mdhip_free(long_list);
  }
  if (ids) mdhip_free(ids);
  mdhip_free(keys);
  return rc;
}

// order-preserving rounds (see k_bid / k_apply)
template <class T, int MODE>
static int scatter_ordered(const mdhip_index_plan *pl, int64_t total, void *dst, const ValDesc &v, T s) {
  // (rounds = the largest multiplicity of a destination, found by reading a flag back per round: not recordable)
  if (md_capturing()) return md_fail(MDHIP_ERUNTIME, "scatter: element-wise duplicates at this size need host-driven rounds and cannot be captured into a graph");
  hipStream_t st = md_stream();
  const int grid = md_grid_for(total);
  const int64_t nslots = total * 2 + 1;
  void *offs = nullptr, *done = nullptr, *owner = nullptr, *flag = nullptr;
  MD_TRY(mdhip_alloc((size_t)total * 8, &offs));
  MD_TRY(mdhip_alloc((size_t)total, &done));
  MD_TRY(mdhip_alloc((size_t)nslots * 8, &owner));
  MD_TRY(mdhip_alloc(sizeof(int), &flag));
  int rc = MDHIP_OK;
  k_offsets<<<grid, MD_BLOCK, 0, st>>>(*pl, total, (int64_t *)offs);
  (void)hipMemsetAsync(done, 0, (size_t)total, st);
  for (int64_t round = 0; round <= total; ++round) {
    (void)hipMemsetAsync(owner, 0xff, (size_t)nslots * 8, st);
    (void)hipMemsetAsync(flag, 0, sizeof(int), st);
    k_bid<<<grid, MD_BLOCK, 0, st>>>((const int64_t *)offs, (const uint8_t *)done, total, (unsigned long long *)owner, nslots);
    k_apply<T, MODE><<<grid, MD_BLOCK, 0, st>>>(*pl, (const int64_t *)offs, (uint8_t *)done, total, (const unsigned long long *)owner,
                                                nslots, (T *)dst, v, s, (int *)flag);
    int remaining = 0;
    rc = read_flag((int *)flag, &remaining);
    if (rc != MDHIP_OK || !remaining) break;
  }
  if (rc == MDHIP_OK) rc = MD_LAUNCH_CHECK("scatter(ordered)");
  mdhip_free(offs); mdhip_free(done); mdhip_free(owner); mdhip_free(flag);
  return rc;
}

// SMALL: a 1- / 2-byte storage type — no atomics of that width and no row-vector path: the ordered kernels serve both modes
template <class T, bool SMALL = false>
static int scatter_typed(const mdhip_index_plan *pl, int64_t total, void *dst, const mdhip_array *val, int mode, bool unique_rows, const int *guard) {
  hipStream_t st = md_stream();
  ValDesc v;
  v.guard = guard;
  v.p = val->data;
  v.is_scalar = val->is_scalar;
  for (int d = 0; d < MDHIP_MAX_NDIM; ++d) v.strides[d] = (!val->is_scalar && d < pl->ndim) ? val->strides[d] : 0;
  T s = val->is_scalar ? md_scalar_as<T>(val) : T();
  constexpr bool is_fp = md_is_float<T>::value;
  // a few positions: walk them in order on one lane — exactly NumPy's loop, one launch. (Up to round 4 this served 4096 positions:
  // ~0.9 us per position of dependent loads, 1.8 ms for the 2000-element put_along_axis of a max / min backward; the sorted paths
  // below cost ~70 us of small launches whatever the count.)
  if (total <= 128) {
    if (mode == MDHIP_SCATTER_SET) k_scatter_serial<T, MDHIP_SCATTER_SET><<<1, 64, 0, st>>>(*pl, total, (T *)dst, v, s);
    else k_scatter_serial<T, MDHIP_SCATTER_ADD><<<1, 64, 0, st>>>(*pl, total, (T *)dst, v, s);
    return MD_LAUNCH_CHECK("scatter(serial)");
  }
  if constexpr (SMALL) {
    // whole rows (the last axis a contiguous run no index varies along): order at ROW granularity, as for the wide types
    int64_t Ls = 0, Ps = 0;
    if (run_geometry(pl, &Ls, &Ps)) {
      if (mode == MDHIP_SCATTER_SET) return scatter_runs<T, MDHIP_SCATTER_SET, false>(pl, Ps, Ls, dst, v, s, false);
      return scatter_runs<T, MDHIP_SCATTER_ADD, false>(pl, Ps, Ls, dst, v, s, false);
    }
    if (mode == MDHIP_SCATTER_SET) return scatter_sorted<T, MDHIP_SCATTER_SET>(pl, total, dst, v, s);
    return scatter_sorted<T, MDHIP_SCATTER_ADD>(pl, total, dst, v, s);
  } else {
  int64_t L = 0, P = 0;
  const bool runs = run_geometry(pl, &L, &P);
  if (mode == MDHIP_SCATTER_SET) {
    if (runs) return scatter_runs<T, MDHIP_SCATTER_SET>(pl, P, L, dst, v, s, unique_rows);
    return scatter_sorted<T, MDHIP_SCATTER_SET>(pl, total, dst, v, s);
  }
  if constexpr (md_same<T, uint8_t>::value) {
    return scatter_sorted<T, MDHIP_SCATTER_ADD>(pl, total, dst, v, s);   // (bool: logical or per destination)
  } else if constexpr (!is_fp) {
    k_scatter_add_int<T><<<md_grid_for(total), MD_BLOCK, 0, st>>>(*pl, total, (T *)dst, v, s);
    return MD_LAUNCH_CHECK("scatter(add,int)");
  } else {
    if (runs) return scatter_runs<T, MDHIP_SCATTER_ADD>(pl, P, L, dst, v, s, unique_rows);
    return scatter_sorted<T, MDHIP_SCATTER_ADD>(pl, total, dst, v, s);
  }
  }
}

// ---- nonzero: per-block counts -> exclusive scan -> ordered compaction ------------------
constexpr int NZ_CHUNK = 2048;  // elements per block (256 threads x 8 consecutive elements)

template <class T>
__global__ void __launch_bounds__(MD_BLOCK) k_nz_count(const T *x, int64_t n, int64_t *block_counts) {
  __shared__ int wsum[MD_BLOCK / 64];
  const int64_t base = (int64_t)blockIdx.x * NZ_CHUNK + (int64_t)threadIdx.x * 8;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (base + j < n && x[base + j] != (T)0) ++c;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (int64_t)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// exclusive scan of the block counts by one block (blocks <= a few 10^5: sequential chunks per lane)
__global__ void __launch_bounds__(MD_BLOCK) k_nz_scan(int64_t *counts, int64_t nb, int64_t *total) {
  __shared__ int64_t part[MD_BLOCK];
  const int64_t per = (nb + MD_BLOCK - 1) / MD_BLOCK;
  const int64_t lo = (int64_t)threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
  int64_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    for (int t = 0; t < MD_BLOCK; ++t) { const int64_t v = part[t]; part[t] = run; run += v; }
    *total = run;
  }
  __syncthreads();
  int64_t run = part[threadIdx.x];
  for (int64_t i = lo; i < hi; ++i) { const int64_t v = counts[i]; counts[i] = run; run += v; }
}
template <class T>
__global__ void __launch_bounds__(MD_BLOCK) k_nz_fill(const T *x, int64_t n, const int64_t *block_offsets, int64_t *out) {
  __shared__ int tcount[MD_BLOCK];
  const int64_t base = (int64_t)blockIdx.x * NZ_CHUNK + (int64_t)threadIdx.x * 8;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (base + j < n && x[base + j] != (T)0) ++c;
  tcount[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {  // 256-entry exclusive scan
    int run = 0;
    for (int t = 0; t < MD_BLOCK; ++t) { const int v = tcount[t]; tcount[t] = run; run += v; }
  }
  __syncthreads();
  int64_t pos = block_offsets[blockIdx.x] + tcount[threadIdx.x];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (base + j < n && x[base + j] != (T)0) out[pos++] = base + j;
}

template <class T> static int nz_run(const mdhip_array *x, int64_t n, int64_t *count_out, int64_t *out_flat) {
  hipStream_t st = md_stream();
  const int64_t nb = (n + NZ_CHUNK - 1) / NZ_CHUNK;
  void *counts = nullptr, *total = nullptr;
  MD_TRY(mdhip_alloc((size_t)(nb + 1) * 8, &counts));
  MD_TRY(mdhip_alloc(8, &total));
  k_nz_count<T><<<(unsigned)nb, MD_BLOCK, 0, st>>>((const T *)x->data, n, (int64_t *)counts);
  k_nz_scan<<<1, MD_BLOCK, 0, st>>>((int64_t *)counts, nb, (int64_t *)total);
  int rc = MD_LAUNCH_CHECK("nonzero(count)");
  if (rc == MDHIP_OK && out_flat) {
    k_nz_fill<T><<<(unsigned)nb, MD_BLOCK, 0, st>>>((const T *)x->data, n, (const int64_t *)counts, out_flat);
    rc = MD_LAUNCH_CHECK("nonzero(fill)");
  }
  if (rc == MDHIP_OK && count_out) {
    rc = md_hip_check(hipMemcpyAsync(count_out, total, 8, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(count)");
    if (rc == MDHIP_OK) rc = md_hip_check(hipStreamSynchronize(st), "hipStreamSynchronize");
  }
  mdhip_free(counts);
  mdhip_free(total);
  return rc;
}
static int nz_dispatch(const mdhip_array *x, int64_t *count_out, int64_t *out_flat) {
  MD_TRY(md_check_array(x, "nonzero x"));
  int64_t n = 1, acc = 1;
  for (int d = x->ndim - 1; d >= 0; --d) {
    if (x->shape[d] != 1 && x->strides[d] != acc) return md_fail(MDHIP_EVALUE, "nonzero: operand must be C-contiguous");
    acc *= x->shape[d];
  }
  n = acc;
  if (n == 0) { if (count_out) *count_out = 0; return MDHIP_OK; }
  if ((n + NZ_CHUNK - 1) / NZ_CHUNK > 0x7fffffff) return md_fail(MDHIP_EVALUE, "nonzero: operand too large");
  switch (x->dtype) {
    case MDHIP_BOOL: return nz_run<uint8_t>(x, n, count_out, out_flat);
    case MDHIP_I32: return nz_run<int32_t>(x, n, count_out, out_flat);
    case MDHIP_I64: return nz_run<int64_t>(x, n, count_out, out_flat);
    case MDHIP_F32: return nz_run<float>(x, n, count_out, out_flat);
    case MDHIP_F64: return nz_run<double>(x, n, count_out, out_flat);
  }
  return md_fail(MDHIP_ETYPE, "nonzero: bad dtype");
}

}  // namespace

namespace {
int scatter_by_dtype(const mdhip_index_plan *pl, int64_t total, void *dst, int dtype, const mdhip_array *val, int mode, bool unique_rows, const int *guard);
}
extern "C" {

int mdhip_nonzero_count(const mdhip_array *x, int64_t *count_out) { return nz_dispatch(x, count_out, nullptr); }
int mdhip_nonzero_fill(const mdhip_array *x, int64_t count, int64_t *out_flat) {
  if (count == 0) return MDHIP_OK;
  if (!out_flat) return md_fail(MDHIP_EVALUE, "nonzero: null output");
  return nz_dispatch(x, nullptr, out_flat);
}

int mdhip_gather(const mdhip_index_plan *pl, const void *src, int dtype, const mdhip_array *out) {
  MD_TRY(md_check_plan(pl));
  MD_TRY(md_check_any_array(out, "gather out"));   // (a mover: any of the twelve dtypes, by element size)
  if (dtype < 0 || dtype >= MDHIP_NUM_ALL_DTYPES || out->dtype != dtype) return md_fail(MDHIP_ETYPE, "gather: bad dtype code %d (out has %d)", dtype, out->dtype);
  if (out->ndim != pl->ndim) return md_fail(MDHIP_EVALUE, "gather: out ndim %d != plan ndim %d", out->ndim, pl->ndim);
  const int64_t total = md_plan_total(pl);
  if (total == 0) return MDHIP_OK;
  MdIter oit;
  memset(&oit, 0, sizeof oit);
  for (int d = 0; d < pl->ndim; ++d) oit.strides[0][d] = out->strides[d];
  // Inside a capture the verdict cannot be read back (and a later replay may see other indices): the kernels — which skip an
  // out-of-range position either way — set the library's sticky word instead, reported at the next synchronisation.
  // (the same deferral on request, MDHIP_INDEX_DEFER=1 / option index_defer: an eager 100-row lookup is 47 us with the read-back, 8 without)
  const bool captured = md_capturing() || md_opt(MD_OPT_INDEX_DEFER) != 0;
  void *flag = captured ? (void *)md_sticky() : nullptr;
  if (!captured) MD_TRY(mdhip_alloc(sizeof(int), &flag));
  hipStream_t st = md_stream();
  if (!captured) (void)hipMemsetAsync(flag, 0, sizeof(int), st);
  const int grid = md_grid_for(total);
  // whole 16-B units when the innermost axis is a contiguous run on both sides
  const int es = (int)md_dtype_size(dtype);
  const int nd = pl->ndim;
  bool runs = nd >= 1 && (es == 1 || es == 4 || es == 8) && total >= 4096;
  const int V = runs ? 16 / es : 1;
  if (runs) {
    runs = pl->src_strides[nd - 1] == 1 && out->strides[nd - 1] == 1 && (pl->shape[nd - 1] % V) == 0 &&
           ((uintptr_t)src & 15) == 0 && ((uintptr_t)out->data & 15) == 0;
    for (int k = 0; runs && k < pl->n_idx; ++k) runs = pl->idx_strides[k][nd - 1] == 0 && (pl->idx_mult[k] % V) == 0;
    for (int d = 0; runs && d < nd - 1; ++d) runs = (pl->src_strides[d] % V) == 0 && (out->strides[d] % V) == 0;
  }
  if (runs) {
    const int vshift = V == 16 ? 4 : V == 4 ? 2 : 1;
    const int64_t run_elems = pl->shape[nd - 1], n_runs = total / run_elems;
    const bool runs_on = md_opt(MD_OPT_GATHER_RUNS) != 0;   // 0: per-unit plan arithmetic (A/B)
    if (runs_on && run_elems / V >= 64 && n_runs >= 64) {
      // a wave per run, up to 16 waves per CU resident (64 KiB of loads in flight per CU)
      int64_t blocks = (n_runs + 3) / 4;
      if (blocks > 1024) blocks = 1024;
      k_gather_runs<<<(unsigned)blocks, MD_BLOCK, 0, st>>>(*pl, n_runs, run_elems, vshift, (const uint4 *)src, (uint4 *)out->data, oit, (int *)flag);
    } else {
      k_gather_vec<<<md_grid_for(total / V), MD_BLOCK, 0, st>>>(*pl, total / V, vshift, (const uint4 *)src, (uint4 *)out->data, oit, (int *)flag);
    }
  } else
  switch (md_dtype_size(dtype)) {
    case 1: k_gather<uint8_t><<<grid, MD_BLOCK, 0, st>>>(*pl, total, (const uint8_t *)src, (uint8_t *)out->data, oit, (int *)flag); break;
    case 2: k_gather<uint16_t><<<grid, MD_BLOCK, 0, st>>>(*pl, total, (const uint16_t *)src, (uint16_t *)out->data, oit, (int *)flag); break;
    case 4: k_gather<uint32_t><<<grid, MD_BLOCK, 0, st>>>(*pl, total, (const uint32_t *)src, (uint32_t *)out->data, oit, (int *)flag); break;
    case 8: k_gather<uint64_t><<<grid, MD_BLOCK, 0, st>>>(*pl, total, (const uint64_t *)src, (uint64_t *)out->data, oit, (int *)flag); break;
    default:
      if (!captured) mdhip_free(flag);
      return md_fail(MDHIP_ETYPE, "gather: bad dtype code %d", dtype);
  }
  int rc = MD_LAUNCH_CHECK("gather");
  if (captured) return rc;
  int bad = 0;
  if (rc == MDHIP_OK) rc = read_flag((int *)flag, &bad);  // NumPy raises IndexError synchronously
  mdhip_free(flag);
  if (rc != MDHIP_OK) return rc;
  if (bad) return md_fail(MDHIP_EINDEX, "index is out of bounds for the indexed axis");
  return MDHIP_OK;
}

int mdhip_scatter(const mdhip_index_plan *pl, void *dst, int dtype, const mdhip_array *val, int mode) {
  MD_TRY(md_check_plan(pl));
  if (!val) return md_fail(MDHIP_EVALUE, "scatter: null value");
  if (!val->is_scalar) MD_TRY(md_check_any_array(val, "scatter val"));
  if (!val->is_scalar && val->dtype != dtype) return md_fail(MDHIP_ETYPE, "scatter: value dtype must match destination");
  if (!val->is_scalar && val->ndim != pl->ndim) return md_fail(MDHIP_EVALUE, "scatter: value ndim mismatch");
  if (mode != MDHIP_SCATTER_SET && mode != MDHIP_SCATTER_ADD) return md_fail(MDHIP_EVALUE, "scatter: bad mode %d", mode);
  const int64_t total = md_plan_total(pl);
  if (total == 0) return MDHIP_OK;
  // bounds first: NumPy raises before touching the destination
  void *flag = nullptr;
  MD_TRY(mdhip_alloc(sizeof(int), &flag));
  (void)hipMemsetAsync(flag, 0, sizeof(int), md_stream());
  void *census = nullptr;
  {
    int64_t L = 0, P = 0, unit = 1;
    if (run_geometry(pl, &L, &P, &unit)) {
      int64_t lo, hi;
      run_key_range(pl, L, &lo, &hi);
      const int64_t nkeys = (hi - lo) / unit + 1;
      const bool census_on = md_opt(MD_OPT_SCATTER_CENSUS) != 0;   // 0: always sort (A/B)
      if (census_on && total > 4096 && nkeys > 0 && nkeys <= (1ll << 22) && mdhip_alloc((size_t)nkeys * sizeof(int), &census) == MDHIP_OK) {
        (void)hipMemsetAsync(census, 0, (size_t)nkeys * sizeof(int), md_stream());
        k_check_bounds_dups<<<md_grid_for(P), MD_BLOCK, 0, md_stream()>>>(*pl, P, L, lo, unit, (int *)census, (int *)flag);
      } else {
        census = nullptr;
        k_check_bounds<<<md_grid_for(P), MD_BLOCK, 0, md_stream()>>>(*pl, P, L, (int *)flag);
      }
    } else {
      k_check_bounds<<<md_grid_for(total), MD_BLOCK, 0, md_stream()>>>(*pl, total, 1, (int *)flag);
    }
  }
  int bits = 0, rc = MDHIP_OK;
  const bool captured = md_capturing() || md_opt(MD_OPT_INDEX_DEFER) != 0;
  const int *guard = nullptr;
  if (captured) {
    // no read-back inside a capture: the writing kernels look at the flag themselves (ValDesc::guard), the sticky word carries
    // the verdict to the next synchronisation, and rows count as possibly repeated (bit 1 of the census is not known here)
    k_flag_to_sticky<<<1, 1, 0, md_stream()>>>((const int *)flag, md_sticky());
    guard = (const int *)flag;
    bits = 2;
  } else {
    rc = read_flag((int *)flag, &bits);
    mdhip_free(flag);
  }
  const bool unique_rows = census != nullptr && (bits & 2) == 0;
  if (census) mdhip_free(census);
  if (rc != MDHIP_OK) return rc;
  if (bits & 1) return md_fail(MDHIP_EINDEX, "index is out of bounds for the indexed axis");
  rc = scatter_by_dtype(pl, total, dst, dtype, val, mode, unique_rows, guard);
  if (captured) mdhip_free(flag);   // (stream-ordered reuse: the graph's own pool keeps the block for later nodes of this capture)
  return rc;
}
}  // extern "C"
namespace {
int scatter_by_dtype(const mdhip_index_plan *pl, int64_t total, void *dst, int dtype, const mdhip_array *val, int mode, bool unique_rows, const int *guard) {
  switch (dtype) {
    case MDHIP_BOOL: return scatter_typed<uint8_t>(pl, total, dst, val, mode, unique_rows, guard);
    case MDHIP_I32: return scatter_typed<int32_t>(pl, total, dst, val, mode, unique_rows, guard);
    case MDHIP_I64: return scatter_typed<int64_t>(pl, total, dst, val, mode, unique_rows, guard);
    case MDHIP_F32: return scatter_typed<float>(pl, total, dst, val, mode, unique_rows, guard);
    case MDHIP_F64: return scatter_typed<double>(pl, total, dst, val, mode, unique_rows, guard);
    // storage-only dtypes in their own type: SET moves bytes; ADD wraps for the integers (uint32 / uint64 add as int32 / int64: the
    // same bits, integer atomics) and rounds to float16 after every contribution, in index order — np.add.at on a half array
    case MDHIP_U32: return scatter_typed<int32_t>(pl, total, dst, val, mode, unique_rows, guard);
    case MDHIP_U64: return scatter_typed<int64_t>(pl, total, dst, val, mode, unique_rows, guard);
    case MDHIP_I8: case MDHIP_U8: return scatter_typed<int8_t, true>(pl, total, dst, val, mode, false, guard);
    case MDHIP_I16: case MDHIP_U16: return scatter_typed<int16_t, true>(pl, total, dst, val, mode, false, guard);
    case MDHIP_F16: return scatter_typed<f16, true>(pl, total, dst, val, mode, false, guard);
  }
  return md_fail(MDHIP_ETYPE, "scatter: bad dtype code %d", dtype);
}

// ---- random permutation of 0..n-1 (opt-in device RNG, md_rng.h): sort the indices by a 64-bit Philox key each --------------------
// (numpy.py:135-136 permutation / shuffle: the caller gathers with the result). The stable LSD radix sort above, 8 passes.
__global__ void __launch_bounds__(MD_BLOCK) k_perm_keys(uint64_t *__restrict__ keys, int64_t *__restrict__ ids, int64_t n, uint64_t seed, uint64_t offset) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
    keys[i] = md_rng_key(seed, offset, i);
    ids[i] = i;
  }
}
}  // namespace
extern "C" {
int mdhip_random_permutation(uint64_t seed, uint64_t offset, const mdhip_array *out) {
  MD_TRY(md_check_array(out, "permutation out"));
  if (out->dtype != MDHIP_I64 || out->ndim != 1 || (out->shape[0] > 1 && out->strides[0] != 1))
    return md_fail(MDHIP_EVALUE, "random_permutation: out must be a contiguous 1-D int64 array");
  const int64_t n = out->shape[0];
  if (n == 0) return MDHIP_OK;
  void *keys = nullptr, *ids = nullptr;
  MD_TRY(mdhip_alloc((size_t)n * 16, &keys));
  int rc = mdhip_alloc((size_t)n * 16, &ids);
  if (rc == MDHIP_OK) {
    k_perm_keys<<<md_grid_for(n), MD_BLOCK, 0, md_stream()>>>((uint64_t *)keys, (int64_t *)ids, n, seed, offset);
    int half = 0;
    rc = radix_sort_pairs((uint64_t *)keys, (int64_t *)ids, n, 64, &half);
    if (rc == MDHIP_OK)
      rc = md_hip_check(hipMemcpyAsync(out->data, (int64_t *)ids + half * n, (size_t)n * 8, hipMemcpyDeviceToDevice, md_stream()), "permutation copy");
  }
  if (ids) mdhip_free(ids);
  mdhip_free(keys);
  return rc;
}
}  // extern "C"
