// reduce_lab.hip — round-3 experiment bench for the single-launch ("last block merges") reductions.
// NOT product code: explores geometry variants of the column sum / full sum / mask product in the cache context of
// the cfg4 backward and prints nothing but a correctness line per variant — the numbers come from
// `rocprofv3 --kernel-trace --stats` of this binary (every variant is a distinct kernel name).
//   build: make -C scripts/lab        run: scripts/lab/run_reduce_lab.sh <tag>
// The surrounding sequence (z > 0, where, the baseline reductions) goes through libmdhip's C-ABI so that the cache state
// in front of each experimental kernel is the product's.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/mdhip.h"

hipStream_t md_stream();  // exported by libmdhip.so

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)
#define MD(x)                                                                 \
  do {                                                                        \
    int s_ = (x);                                                             \
    if (s_ != 0) { fprintf(stderr, "%s:%d mdhip: %s\n", __FILE__, __LINE__, mdhip_last_error()); exit(1); } \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld16(const float *p, int nt) {
  if (nt) {
    i4 t = __builtin_nontemporal_load(reinterpret_cast<const i4 *>(p));
    f4 v;
    __builtin_memcpy(&v, &t, 16);
    return v;
  }
  return *reinterpret_cast<const f4 *>(p);
}
__device__ __forceinline__ void st16(float *p, f4 v, int nt) {
  if (nt) {
    i4 t;
    __builtin_memcpy(&t, &v, 16);
    __builtin_nontemporal_store(t, reinterpret_cast<i4 *>(p));
  } else {
    *reinterpret_cast<f4 *>(p) = v;
  }
}
// write-through (sc1) 16-B store / L1-bypassing (sc1) 16-B load through a buffer descriptor (guide §6 G16, R1)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, f4 v) {
  i4 t;
  __builtin_memcpy(&t, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, 16);
}
__device__ __forceinline__ f4 ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
  i4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
  f4 v;
  __builtin_memcpy(&v, &t, 16);
  return v;
}
// one lane of the block: "my partials are out" -> true in the block that arrived last
__device__ __forceinline__ bool ticket_last(unsigned *ticket, unsigned n_arrivals, unsigned *lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_flag = old == n_arrivals - 1;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// ------------------------------------------------------------------ column sums, 2-D grid ----
struct ColsArgs {
  const float *x;       // MODE 0: the matrix
  const uint8_t *mask;  // MODE 1: bool matrix; the evaluated value is g * mask
  const float *gptr;    // MODE 1: device scalar
  float *y;             // MODE 1: evaluated output
  float *partial;       // [NB][n_cols]
  unsigned *tickets;    // [NS], zero before the launch, zero after it
  float *out;           // [n_cols]
  int n_cols, n_rows, rows_band, NS, NB;
};

// Block = 4 waves arranged WR (along rows) x WC (along columns); a wave reads QB 1-KiB pieces of a row per row;
// two batches of RB rows in flight per wave. Grid = NS column strips x NB row bands; a band's blocks are neighbours.
// TICKET: the block that arrives last at its strip's ticket merges the strip's NB partial rows (fixed order).
// ORDER: 0 bands, rows ascending; 1 the same walked from the LAST row back (the matrix was just written front to back:
// its tail is what the Infinity Cache still holds); 2 rows interleaved over the bands (the grid reads one moving window); 3 = 2 reversed
template <int QB, int WR, int RB, int NT, int MODE, int TICKET, int NBT, int ORDER, int NW = 4>
__global__ void __launch_bounds__(NW * 64) k_cols2d(ColsArgs a) {
  constexpr int WC = NW / WR;
  constexpr int BC = WC * QB * 256;  // floats per block row
  __shared__ __attribute__((aligned(16))) float sm[WR > 1 ? (WR - 1) * WC * QB * 64 : 1][4];
  __shared__ unsigned flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wc = w % WC, wr = w / WC;
  const int s = blockIdx.x % a.NS, b = blockIdx.x / a.NS;
  const int64_t col0 = (int64_t)s * BC + wc * QB * 256 + lane * 4;
  const int64_t r0 = (int64_t)b * a.rows_band;
  int64_t r1 = r0 + a.rows_band;
  if (r1 > a.n_rows) r1 = a.n_rows;
  const int64_t nrw = (r1 - r0 - wr + WR - 1) / WR;  // rows of this wave: r0 + wr + WR * i
  auto rowof = [&](int64_t i) -> int64_t {
    int64_t row = (ORDER & 2) ? (i * WR + wr) * NBT + b : r0 + wr + i * WR;
    return (ORDER & 1) ? a.n_rows - 1 - row : row;
  };
  const int64_t nb = nrw / RB;
  float g = 0.f;
  if (MODE == 1) g = *a.gptr;
  f4 acc[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) acc[q] = (f4){0.f, 0.f, 0.f, 0.f};
  f4 t[2][RB][QB];
  uint32_t m[2][RB][QB];
  auto load = [&](int buf, int64_t bt) {
    const int64_t i0 = (bt < nb ? bt : nb - 1) * RB;
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int64_t row = rowof(i0 + u);
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        if (MODE == 0) t[buf][u][q] = ld16(a.x + row * a.n_cols + col0 + q * 256, NT);
        else m[buf][u][q] = *reinterpret_cast<const uint32_t *>(a.mask + row * a.n_cols + col0 + q * 256);
      }
    }
  };
  auto use = [&](int buf, int64_t bt) {
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int q = 0; q < QB; ++q) {
        if (MODE == 0) acc[q] += t[buf][u][q];
        else {
          const uint32_t mm = m[buf][u][q];
          f4 v;
          v.x = g * (float)(mm & 0xffu);
          v.y = g * (float)((mm >> 8) & 0xffu);
          v.z = g * (float)((mm >> 16) & 0xffu);
          v.w = g * (float)(mm >> 24);
          acc[q] += v;
          const int64_t row = rowof(bt * RB + u);
          st16(a.y + row * a.n_cols + col0 + q * 256, v, 1);
        }
      }
  };
  if (nb > 0) {
    load(0, 0);
    int64_t bt = 0;
    for (; bt + 1 < nb; bt += 2) {
      load(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
      use(0, bt);
      __builtin_amdgcn_sched_barrier(0);
      load(0, bt + 2);
      __builtin_amdgcn_sched_barrier(0);
      use(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bt < nb) use(0, bt);
  }
  for (int64_t i = nb * RB; i < nrw; ++i) {
    const int64_t row = rowof(i);
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      if (MODE == 0) acc[q] += ld16(a.x + row * a.n_cols + col0 + q * 256, 0);
      else {
        const uint32_t mm = *reinterpret_cast<const uint32_t *>(a.mask + row * a.n_cols + col0 + q * 256);
        f4 v;
        v.x = g * (float)(mm & 0xffu);
        v.y = g * (float)((mm >> 8) & 0xffu);
        v.z = g * (float)((mm >> 16) & 0xffu);
        v.w = g * (float)(mm >> 24);
        acc[q] += v;
        st16(a.y + row * a.n_cols + col0 + q * 256, v, 1);
      }
    }
  }
  // wave rows -> wave row 0, in wave-row order
  if (WR > 1) {
    if (wr > 0) {
#pragma unroll
      for (int q = 0; q < QB; ++q) *reinterpret_cast<f4 *>(sm[(((wr - 1) * WC + wc) * QB + q) * 64 + lane]) = acc[q];
    }
    __syncthreads();
    if (wr == 0) {
#pragma unroll
      for (int k = 0; k < WR - 1; ++k)
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] += *reinterpret_cast<const f4 *>(sm[((k * WC + wc) * QB + q) * 64 + lane]);
    }
  }
  const __amdgpu_buffer_rsrc_t pr = rsrc_of(a.partial, (unsigned)((size_t)a.NB * a.n_cols * 4));
  if (wr == 0) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      const unsigned off = (unsigned)(((int64_t)b * a.n_cols + col0 + q * 256) * 4);
      if (TICKET) st16_sc1(pr, off, acc[q]);
      else *reinterpret_cast<f4 *>(a.partial + (int64_t)b * a.n_cols + col0 + q * 256) = acc[q];
    }
  }
  if (!TICKET) return;
  if (!ticket_last(a.tickets + s, (unsigned)a.NB, &flag)) return;
  // merge: wave w takes partial rows w, w+4, ..; the strip's BC columns = BC/256 pieces of 64 lanes x 4
  constexpr int PC = BC / 256;
  f4 tot[PC];
#pragma unroll
  for (int p = 0; p < PC; ++p) tot[p] = (f4){0.f, 0.f, 0.f, 0.f};
  for (int r = w; r < a.NB; r += 4 * NW) {  // four rows of this wave in flight
    f4 v[4][PC];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < PC; ++p) {
        const int rr = r + NW * k;
        v[k][p] = rr < a.NB ? ld16_sc1(pr, (unsigned)(((int64_t)rr * a.n_cols + (int64_t)s * BC + p * 256 + lane * 4) * 4)) : (f4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int p = 0; p < PC; ++p) tot[p] += v[k][p];
  }
  __shared__ __attribute__((aligned(16))) float sm2[(NW - 1) * PC * 64][4];
  if (w > 0) {
#pragma unroll
    for (int p = 0; p < PC; ++p) *reinterpret_cast<f4 *>(sm2[((w - 1) * PC + p) * 64 + lane]) = tot[p];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int p = 0; p < PC; ++p) {
#pragma unroll
      for (int k = 0; k < NW - 1; ++k) tot[p] += *reinterpret_cast<const f4 *>(sm2[(k * PC + p) * 64 + lane]);
      *reinterpret_cast<f4 *>(a.out + (int64_t)s * BC + p * 256 + lane * 4) = tot[p];
    }
  }
  if (tid == 0) __hip_atomic_store(a.tickets + s, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// second launch of the two-launch form: partial [NB][n_cols] -> out (the product's k_reduce_cols_merge shape)
__global__ void __launch_bounds__(1024) k_cols_merge(const float *__restrict__ partial, int n_cols, int n_rows, float *__restrict__ out) {
  __shared__ float smem[15][64];
  const int cx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cx;
  float acc = 0.f;
  float t[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int r = w + 16 * u;
    t[u] = r < n_rows ? partial[(int64_t)r * n_cols + col] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) acc += t[u];
  if (w > 0) smem[w - 1][cx] = acc;
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < 15; ++k) acc += smem[k][cx];
    out[col] = acc;
  }
}

// ------------------------------------------------------------------ full sum ----
// grid-strided 16-B loads, U in flight per lane; TICKET: last block sums the per-block partials in index order
// TICKET: 0 two launches, 1 one counter, 2 two levels (32 shard counters on lines of their own, then a top counter)
template <int U, int NT, int TICKET, int GRID, int REV>
__global__ void __launch_bounds__(256) k_sum_all(const float *__restrict__ x0, int64_t nvec, float *partial, unsigned *ticket, float *out) {
  const float *x = REV ? x0 + 4 * (nvec - 1) : x0;
  constexpr int SG = REV ? -1 : 1;
  __shared__ float smem[4];
  __shared__ unsigned flag;
  const int64_t gs = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  f4 a[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = (f4){0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * gs < nvec; i += U * gs) {
    f4 t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = ld16(x + SG * 4 * (i + u * gs), NT);
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] += t[u];
  }
  for (; i < nvec; i += gs) a[0] += ld16(x + SG * 4 * i, 0);
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) acc += (a[u].x + a[u].y) + (a[u].z + a[u].w);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    if (TICKET) __hip_atomic_store(partial + blockIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partial[blockIdx.x] = v;
  }
  if (!TICKET) return;
  if (TICKET == 1) {
    if (!ticket_last(ticket, gridDim.x, &flag)) return;
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      unsigned last = 0;
      const unsigned sh = blockIdx.x & 31;
      if (__hip_atomic_fetch_add(ticket + 32 * sh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GRID / 32 - 1) {
        __hip_atomic_store(ticket + 32 * sh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(ticket + 32 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 31;
      }
      flag = last;
    }
    __syncthreads();
    if (!flag) return;
    ticket += 32 * 32;
  }
  float s = 0.f;
  float v[8];
  // <= 2048 partials: 8 per lane, all loads ahead of the adds
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const unsigned idx = threadIdx.x + 256 * k;
    v[k] = idx < gridDim.x ? __hip_atomic_load(partial + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  __syncthreads();
  if (lane == 0) smem[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    *out = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// software-pipelined full sum: two batches of U grid-strided vectors in flight per lane
template <int U, int TICKET, int GRID>
__global__ void __launch_bounds__(256) k_sum_pipe(const float *__restrict__ x, int64_t nvec, float *partial, unsigned *ticket, float *out) {
  __shared__ float smem[4];
  __shared__ unsigned flag;
  const int64_t gs = (int64_t)GRID * 256;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ntrip = nvec / (gs * U);   // whole batches (every lane has all U vectors)
  f4 a[U];
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = (f4){0.f, 0.f, 0.f, 0.f};
  f4 t[2][U];
  auto load = [&](int buf, int64_t bt) {
    const int64_t b = bt < ntrip ? bt : ntrip - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) t[buf][u] = ld16(x + 4 * (i0 + (b * U + u) * gs), 0);
  };
  auto add = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] += t[buf][u];
  };
  if (ntrip > 0) {
    load(0, 0);
    int64_t bt = 0;
    for (; bt + 1 < ntrip; bt += 2) {
      load(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
      add(0);
      __builtin_amdgcn_sched_barrier(0);
      load(0, bt + 2);
      __builtin_amdgcn_sched_barrier(0);
      add(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bt < ntrip) add(0);
  }
  for (int64_t i = i0 + ntrip * U * gs; i < nvec; i += gs) a[0] += ld16(x + 4 * i, 0);
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) acc += (a[u].x + a[u].y) + (a[u].z + a[u].w);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem[w] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    if (TICKET) __hip_atomic_store(partial + blockIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else partial[blockIdx.x] = v;
  }
  if (!TICKET) return;
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      unsigned last = 0;
      const unsigned sh = blockIdx.x & 31;
      if (__hip_atomic_fetch_add(ticket + 32 * sh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == GRID / 32 - 1) {
        __hip_atomic_store(ticket + 32 * sh, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(ticket + 32 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 31;
      }
      flag = last;
    }
    __syncthreads();
    if (!flag) return;
    ticket += 32 * 32;
  }
  float s = 0.f;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const unsigned idx = threadIdx.x + 256 * k;
    v[k] = idx < GRID ? __hip_atomic_load(partial + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  __syncthreads();
  if (lane == 0) smem[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    *out = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void __launch_bounds__(256) k_sum_finish(const float *partial, int n, float *out) {
  __shared__ float smem[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (smem[0] + smem[1]) + (smem[2] + smem[3]);
}

// ------------------------------------------------------------------ mask product ----
// y = g * mask (mask: bool bytes, 4 per lane and vector), U vectors per trip, grid-strided
template <int U, int NTST>
__global__ void __launch_bounds__(256) k_maskprod(const uint8_t *__restrict__ mask, const float *gptr, float *__restrict__ y, int64_t nvec) {
  const float g = *gptr;
  const int64_t gs = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * gs < nvec; i += U * gs) {
    uint32_t m[U];
#pragma unroll
    for (int u = 0; u < U; ++u) m[u] = *reinterpret_cast<const uint32_t *>(mask + 4 * (i + u * gs));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      f4 v;
      v.x = g * (float)(m[u] & 0xffu);
      v.y = g * (float)((m[u] >> 8) & 0xffu);
      v.z = g * (float)((m[u] >> 16) & 0xffu);
      v.w = g * (float)(m[u] >> 24);
      st16(y + 4 * (i + u * gs), v, NTST);
    }
  }
  for (; i < nvec; i += gs) {
    const uint32_t mm = *reinterpret_cast<const uint32_t *>(mask + 4 * i);
    f4 v;
    v.x = g * (float)(mm & 0xffu);
    v.y = g * (float)((mm >> 8) & 0xffu);
    v.z = g * (float)((mm >> 16) & 0xffu);
    v.w = g * (float)(mm >> 24);
    st16(y + 4 * i, v, NTST);
  }
}
// a lane converts 16 mask bytes (one 16-B load) into four 16-B stores: fewer, wider mask loads
template <int NTST>
__global__ void __launch_bounds__(256) k_maskprod16(const uint8_t *__restrict__ mask, const float *gptr, float *__restrict__ y, int64_t n16) {
  const float g = *gptr;
  const int64_t gs = (int64_t)gridDim.x * 256;
  // a wave covers 64 x 16 mask bytes = 1 KiB of mask = 4 KiB of output; store q of lane l goes to vector 4*l + q
  // (uncoalesced across q) -> instead let the wave transpose through the lane mapping: lane l handles mask bytes
  // [16 l, 16 l + 16) and writes y[16 l .. 16 l + 15] as four consecutive 16-B vectors = 64 contiguous bytes per lane.
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += gs) {
    const uint4 mm = *reinterpret_cast<const uint4 *>(mask + 16 * i);
    const uint32_t w4[4] = {mm.x, mm.y, mm.z, mm.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f4 v;
      v.x = g * (float)(w4[q] & 0xffu);
      v.y = g * (float)((w4[q] >> 8) & 0xffu);
      v.z = g * (float)((w4[q] >> 16) & 0xffu);
      v.w = g * (float)(w4[q] >> 24);
      st16(y + 16 * i + 4 * q, v, NTST);
    }
  }
}

// ------------------------------------------------------------------ host ----
static mdhip_array arr2(void *p, int dt, int64_t r, int64_t c) {
  mdhip_array a;
  memset(&a, 0, sizeof a);
  a.data = p; a.dtype = dt; a.ndim = 2;
  a.shape[0] = r; a.shape[1] = c; a.strides[0] = c; a.strides[1] = 1;
  return a;
}
static mdhip_array scal(double v) {
  mdhip_array a;
  memset(&a, 0, sizeof a);
  a.dtype = MDHIP_F32; a.is_scalar = 1; a.scalar_f = v;
  return a;
}

int main(int argc, char **argv) {
  const int R = 8192, Cn = 4096, iters = argc > 1 ? atoi(argv[1]) : 12;
  const int64_t N = (int64_t)R * Cn;
  MD(mdhip_init(0));
  hipStream_t st = md_stream();
  float *z, *r, *gm, *gm2, *out, *out2, *partial, *loss, *gdev;
  uint8_t *mask;
  unsigned *tickets;
  MD(mdhip_alloc(N * 4, (void **)&z)); MD(mdhip_alloc(N * 4, (void **)&r)); MD(mdhip_alloc(N * 4, (void **)&gm)); MD(mdhip_alloc(N * 4, (void **)&gm2));
  MD(mdhip_alloc(N, (void **)&mask));
  MD(mdhip_alloc(Cn * 4, (void **)&out)); MD(mdhip_alloc(Cn * 4, (void **)&out2));
  MD(mdhip_alloc((size_t)1024 * Cn * 4, (void **)&partial));
  MD(mdhip_alloc(64, (void **)&loss)); MD(mdhip_alloc(64, (void **)&gdev));
  CK(hipMalloc(&tickets, 16384)); CK(hipMemset(tickets, 0, 16384));
  {
    std::vector<float> h(N);
    uint64_t s = 88172645463325252ull;
    for (int64_t i = 0; i < N; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((int64_t)(s >> 40) - (1 << 23)) / (float)(1 << 22); }
    CK(hipMemcpy(z, h.data(), N * 4, hipMemcpyHostToDevice));
    const float one = 1.0f;
    CK(hipMemcpy(gdev, &one, 4, hipMemcpyHostToDevice));
  }
  mdhip_array Z = arr2(z, MDHIP_F32, R, Cn), Rr = arr2(r, MDHIP_F32, R, Cn), M = arr2(mask, MDHIP_BOOL, R, Cn), GM = arr2(gm, MDHIP_F32, R, Cn);
  mdhip_array zero = scal(0.0);
  mdhip_array G = arr2(gdev, MDHIP_F32, R, Cn);
  G.strides[0] = G.strides[1] = 0;  // the stride-0 seed view of backward()
  mdhip_array OUT = arr2(out, MDHIP_F32, 1, Cn), LOSS = arr2(loss, MDHIP_F32, 1, 1);
  auto greater = [&] { MD(mdhip_binary(MDHIP_B_GT, &Z, &zero, &M, MDHIP_F32)); };
  auto forward = [&] { greater(); MD(mdhip_where(&M, &Z, &zero, &Rr)); };  // z > 0 ; where(mask, z, 0)
  auto base_loss = [&] { MD(mdhip_reduce(MDHIP_R_SUM, &Rr, &LOSS, 3u)); };
  auto base_maskprod = [&] { MD(mdhip_binary(MDHIP_B_MUL, &G, &M, &GM, MDHIP_F32)); };
  auto base_colsum = [&] { MD(mdhip_reduce(MDHIP_R_SUM, &GM, &OUT, 1u)); };

  std::vector<float> ref(Cn), got(Cn), got2(Cn);
  float ref_loss = 0.f;
  forward(); base_loss(); base_maskprod(); base_colsum();
  MD(mdhip_sync());
  CK(hipMemcpy(ref.data(), out, Cn * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&ref_loss, loss, 4, hipMemcpyDeviceToHost));
  printf("baseline: loss %.6e colsum[0] %.6e\n", ref_loss, ref[0]);
  auto check = [&](const char *name, float *dev, bool second) {
    CK(hipStreamSynchronize(st));
    std::vector<float> &g = second ? got2 : got;
    CK(hipMemcpy(g.data(), dev, Cn * 4, hipMemcpyDeviceToHost));
    double e = 0, m = 0;
    for (int i = 0; i < Cn; ++i) { e = fmax(e, fabs((double)g[i] - ref[i])); m = fmax(m, fabs((double)ref[i])); }
    if (second) printf("  %-52s rel err %.2e  %s\n", name, e / m, memcmp(got.data(), got2.data(), Cn * 4) == 0 ? "bit-identical run to run" : "DIFFERS RUN TO RUN");
  };
  const char *sel = argc > 2 ? argv[2] : "all";
  auto want = [&](const char *grp) { return !strcmp(sel, "all") || strstr(sel, grp); };

  // ---- E: column sum (eager context: forward, loss, mask product, column sum; lazy context: z > 0, evalcols) ----
#define COLSX(QB, WR, RB, NT, MODE, TK, NBv, ORD, NW, CTX)                                                          \
  {                                                                                                            \
    char label[128];                                                                                           \
    snprintf(label, sizeof label, "%s<QB%d WR%d RB%d NT%d TK%d NB%d ORD%d NW%d> ctx%d", MODE ? "evalcols" : "cols", QB, WR, RB, NT, TK, NBv, ORD, NW, CTX); \
    ColsArgs a;                                                                                                \
    a.x = gm; a.mask = mask; a.gptr = gdev; a.y = gm2; a.partial = partial; a.tickets = tickets; a.out = out2; \
    a.n_cols = Cn; a.n_rows = R; a.NS = Cn / ((NW / WR) * QB * 256); a.NB = NBv; a.rows_band = R / (NBv);     \
    if (a.NS * ((NW / WR) * QB * 256) != Cn || R % (NBv) != 0) { printf("bad geometry %s\n", label); exit(1); } \
    for (int it = 0; it < iters; ++it) {                                                                       \
      if (MODE == 0) { forward(); base_loss(); if (CTX == 0) base_maskprod(); else if (CTX == 1) k_maskprod<2, 1><<<1024, 256, 0, st>>>(mask, gdev, gm, N / 4); else k_maskprod<2, 0><<<1024, 256, 0, st>>>(mask, gdev, gm, N / 4); } \
      else greater();                                                                                          \
      CK(hipMemsetAsync(out2, 0xff, Cn * 4, st));                                                              \
      k_cols2d<QB, WR, RB, NT, MODE, TK, NBv, ORD, NW><<<a.NS * a.NB, NW * 64, 0, st>>>(a);                     \
      if (!TK) k_cols_merge<<<Cn / 64, 1024, 0, st>>>(partial, Cn, a.NB, out2);                                 \
      if (it < 2) check(label, out2, it == 1);                                                                 \
    }                                                                                                          \
  }
#define COLS(QB, WR, RB, NT, MODE, TK, NBv, ORD, NW) COLSX(QB, WR, RB, NT, MODE, TK, NBv, ORD, NW, 0)
  if (want("base")) {
    for (int it = 0; it < iters; ++it) { forward(); base_loss(); base_maskprod(); base_colsum(); }
    CK(hipStreamSynchronize(st));
  }
  if (want("cols")) {
    COLS(4, 1, 3, 0, 0, 0, 256, 0, 4);  // replica of the product's two-launch sweep
    COLS(1, 4, 8, 0, 0, 1, 16, 0, 4);
    COLS(1, 4, 8, 0, 0, 1, 16, 2, 4);
    COLS(1, 4, 4, 0, 0, 1, 16, 2, 4);
    COLS(1, 4, 6, 0, 0, 1, 16, 2, 4);
    COLS(1, 4, 12, 0, 0, 1, 16, 2, 4);
    COLS(1, 4, 8, 0, 0, 1, 8, 2, 4);
    COLS(1, 4, 8, 0, 0, 1, 32, 2, 4);
    COLS(1, 4, 4, 0, 0, 1, 32, 2, 4);
    COLS(1, 4, 4, 0, 0, 1, 64, 2, 4);
    COLS(1, 8, 4, 0, 0, 1, 16, 2, 8);
    COLS(1, 8, 6, 0, 0, 1, 16, 2, 8);
    COLS(1, 8, 8, 0, 0, 1, 8, 2, 8);
    COLS(1, 16, 4, 0, 0, 1, 8, 2, 16);
    COLS(1, 2, 8, 0, 0, 1, 16, 2, 4);
    COLS(2, 4, 4, 0, 0, 1, 32, 2, 4);
    COLS(1, 4, 8, 1, 0, 1, 16, 2, 4);
    COLS(1, 4, 8, 0, 0, 0, 16, 2, 4);   // interleaved, two launches (what the ticket costs)
    COLSX(1, 4, 8, 0, 0, 1, 16, 2, 4, 1);  // after a mask product with non-temporal stores
    COLSX(1, 4, 8, 0, 0, 1, 16, 2, 4, 2);  // after the lab's mask product with plain stores
  }
  if (want("evalcols")) {
    COLS(4, 1, 2, 0, 1, 0, 256, 0, 4);  // replica of the generated two-launch sweep
    COLS(1, 4, 8, 0, 1, 1, 16, 0, 4);
    COLS(1, 4, 8, 0, 1, 1, 16, 2, 4);
    COLS(1, 4, 4, 0, 1, 1, 16, 2, 4);
    COLS(1, 4, 2, 0, 1, 1, 16, 2, 4);
    COLS(1, 4, 4, 0, 1, 1, 32, 2, 4);
    COLS(1, 4, 4, 0, 1, 1, 64, 2, 4);
    COLS(1, 8, 4, 0, 1, 1, 16, 2, 8);
    COLS(1, 8, 2, 0, 1, 1, 32, 2, 8);
    COLS(1, 16, 2, 0, 1, 1, 16, 2, 16);
  }
  // ---- C: the loss sum (context: forward just wrote r) ----
#define SUMALL(U, NT, TK, GRID, REV)                                                                   \
  {                                                                                                    \
    char label[128];                                                                                   \
    snprintf(label, sizeof label, "sum<U%d NT%d TK%d GRID%d REV%d>", U, NT, TK, GRID, REV);            \
    for (int it = 0; it < iters; ++it) {                                                               \
      forward();                                                                                       \
      k_sum_all<U, NT, TK, GRID, REV><<<GRID, 256, 0, st>>>(r, N / 4, partial, tickets + 2048, loss);   \
      if (!TK) k_sum_finish<<<1, 256, 0, st>>>(partial, GRID, loss);                                   \
      if (it == 0) {                                                                                   \
        CK(hipStreamSynchronize(st));                                                                  \
        float v;                                                                                       \
        CK(hipMemcpy(&v, loss, 4, hipMemcpyDeviceToHost));                                             \
        printf("  %-52s rel err %.2e\n", label, fabs(v - ref_loss) / fabs(ref_loss));                 \
      }                                                                                                \
    }                                                                                                  \
  }
#define SUMPIPE(U, TK, GRID)                                                                           \
  {                                                                                                    \
    char label[128];                                                                                   \
    snprintf(label, sizeof label, "sumpipe<U%d TK%d GRID%d>", U, TK, GRID);                            \
    for (int it = 0; it < iters; ++it) {                                                               \
      forward();                                                                                       \
      k_sum_pipe<U, TK, GRID><<<GRID, 256, 0, st>>>(r, N / 4, partial, tickets + 2048, loss);           \
      if (!TK) k_sum_finish<<<1, 256, 0, st>>>(partial, GRID, loss);                                   \
      if (it == 0) {                                                                                   \
        CK(hipStreamSynchronize(st));                                                                  \
        float v;                                                                                       \
        CK(hipMemcpy(&v, loss, 4, hipMemcpyDeviceToHost));                                             \
        printf("  %-52s rel err %.2e\n", label, fabs(v - ref_loss) / fabs(ref_loss));                 \
      }                                                                                                \
    }                                                                                                  \
  }
  if (want("sum")) {
    for (int it = 0; it < iters; ++it) { forward(); base_loss(); }
    SUMALL(2, 0, 0, 1024, 0);
    SUMALL(2, 0, 2, 1024, 0);
    SUMPIPE(2, 0, 1024);
    SUMPIPE(2, 1, 1024);
    SUMPIPE(4, 1, 1024);
    SUMPIPE(4, 1, 512);
    SUMPIPE(8, 1, 512);
    SUMPIPE(8, 1, 256);
    SUMPIPE(4, 1, 256);
    SUMPIPE(16, 1, 256);
    SUMPIPE(1, 1, 1024);
    SUMPIPE(1, 1, 2048);
  }
  // ---- D: the mask product (context: forward + loss sum) ----
#define MASKP(U, NTST, GRID)                                                       \
  {                                                                                \
    char label[128];                                                               \
    snprintf(label, sizeof label, "maskprod<U%d NTST%d> grid %d", U, NTST, GRID);  \
    for (int it = 0; it < iters; ++it) {                                           \
      forward(); base_loss();                                                      \
      k_maskprod<U, NTST><<<GRID, 256, 0, st>>>(mask, gdev, gm, N / 4);             \
      base_colsum();                                                               \
      if (it == 0) check(label, out, false), check(label, out, true);              \
    }                                                                              \
  }
  if (want("maskprod")) {
    MASKP(2, 0, 1024);
    MASKP(4, 0, 1024);
    MASKP(2, 1, 1024);
  }
  CK(hipStreamSynchronize(st));
  printf("done\n");
  return 0;
}
