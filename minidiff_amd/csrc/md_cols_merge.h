// md_cols_merge.h — second pass of the sweep-style column reductions (reduce.hip: k_reduce_cols_sweep;
// fusion_jit.inc: the generated sweep kernels). Included INSIDE each translation unit's anonymous namespace.
#pragma once
// Merge of the sweep kernel's partial rows (one per block: <= 256 of them). The tiled kernel above took 6.5 us for
// this 4 MiB (16 blocks, 16 dependent load rounds per wave); here a block is 64 columns x 16 row waves (16 x LOADS partial rows), every wave
// issues the loads of ALL its rows before the first add (one round trip) and the waves are combined through LDS in
// wave order, so the sum order is fixed: row w, w+16, w+32, .. inside wave w, then waves 0..15.
template <class R, class Tacc, class To, int LOADS = 16>
__global__ void __launch_bounds__(1024) k_reduce_cols_merge(const Tacc *__restrict__ partial, int64_t n_out, int64_t n_rows, To *__restrict__ out) {
  __shared__ Tacc smem[15][64];
  const int cx = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + cx;
  Tacc acc = R::template identity<Tacc>();
  if (col < n_out) {
    Tacc t[LOADS];
#pragma unroll
    for (int u = 0; u < LOADS; ++u) {
      const int64_t r = w + 16 * u;
      t[u] = r < n_rows ? partial[r * n_out + col] : R::template identity<Tacc>();
    }
#pragma unroll
    for (int u = 0; u < LOADS; ++u) acc = R::combine(acc, t[u]);
  }
  if (w > 0) smem[w - 1][cx] = acc;
  __syncthreads();
  if (w == 0 && col < n_out) {
#pragma unroll
    for (int k = 0; k < 15; ++k) acc = R::combine(acc, smem[k][cx]);
    out[col] = md_cast<To>(acc);
  }
}
