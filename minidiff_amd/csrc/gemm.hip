// gemm.hip — matmul forward/backward on gfx950 matrix cores.
//
// Serves reference minidiff/backend/numpy.py:84 (np.matmul; also dot/tensordot
// :68,91 through the shim) and with it the three GEMMs of
// minidiff/ops/definitions.py:487-492:  C = A·B (NN), dA = G·Bᵀ (NT),
// dB = Aᵀ·G (TN). The transposes arrive as strided VIEWS (x.T is a stride
// permutation), so the kernel takes element strides for both operands and picks
// a tile loader per layout instead of materialising a transpose.
//
// f32: v_mfma_f32_32x32x2_f32 — exact f32 fma chain in k order, 256 FLOP/clk/CU
// (MI355X_MICROARCH.md "Matrix cores"): bound = 157.3 TFLOP/s.
//   block tile BM x BN x BK (256x128x16 for large problems; a cost model picks smaller tiles
//   for grids that do not fill whole rounds), 4 waves (2x2; 8 on the 128x128 tile of one-tile-per-CU grids), each wave (BM/2)x(BN/2) in 32x32
//   MFMA tiles (128 accumulator VGPRs at 256x128). Operands are staged in LDS k-major ([k][m], [k][n])
//   so a fragment read is one conflict-free ds_read_b32 per MFMA operand. Three
//   tiles are in flight per block: LDS buffer `cur` (being multiplied), the other
//   LDS buffer (being written from registers during the first MFMA steps) and the
//   registers (global loads of tile k+2 issued in step 2); fragment reads run one
//   step ahead in a register double buffer; one barrier per k-step. The k-tile body is
//   branch-free (one scheduling region) and sched_group_barrier places a slice of the step's
//   LDS reads / LDS writes / global loads behind EVERY MFMA, so a wave keeps the matrix pipe
//   fed while its memory work issues in the gaps (136 -> 139.5 TFLOP/s at 4096^3; with one
//   block per CU 60 -> 66). The loop is rotated: a tile's first fragments are read right
//   behind the barrier that publishes it, under the previous step's MFMAs.
//   Workgroup ids are remapped so the 8 XCDs each own a contiguous band of
//   output tiles (per-XCD L2 locality on the shared A row panel).
// Aligned operands whose tiles divide the problem (every BASELINE shape) run the DIRECT-TO-LDS kernels further down
// (k_gemm_f32_kc_glds for NN / NT, k_gemm_f32_tn_glds for TN: global_load_lds_dwordx4, no staging registers, no ds_write;
// 256x256x32 tiles, or eight waves on one 128x128x32 tile for grids of one tile per CU; DMA addresses as scalar base + 32-bit
// lane offset): 150-152 TFLOP/s at 4096^3 against 139.5 for the register-staged kernel described above, which keeps the
// unaligned / 16-deep / split-K cases (ragged sizes with 4-multiples run RAGGED variants of the direct-to-LDS kernels).
// ragged or unaligned shapes: guarded edge variant of the same kernel; few tiles and a long k:
// split-K with a deterministic split-order sum. f64: the same scheme on v_mfma_f64_16x16x4_f64
// (k_gemm_f64_mfma). Integers / tiny problems: a plain LDS-tiled kernel.
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include <hip/hip_ext.h>

#include "md_hip.h"

extern "C" int mdhip_alloc(size_t, void **);
extern "C" int mdhip_unary(int, const mdhip_array *, const mdhip_array *);
extern "C" int mdhip_free(void *);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LDP = 4;  // LDS row pad (floats): keeps the transposing ds_write_b32 at <=2-way conflicts

struct GemmArgs {
  const float *A, *B;
  float *C;
  int64_t M, N, K;
  int64_t a_bs, a_ms, a_ks, b_bs, b_ks, b_ns, c_bs, c_ms, c_ns;
  int tiles_m, tiles_n;
  int super_h;  // tile rows per band of the XCD-compact numbering (0: plain row-major)
  int vec_ok;  // EDGE kernels: operands are 16-B aligned, so a fully inside vector may be loaded whole
  // split-K (gridDim.y > 1): block y multiplies k in [y*k_chunk, (y+1)*k_chunk) into its own
  // partial C (C + y*c_split), summed afterwards in split order by k_gemm_splitk_sum
  int64_t k_chunk, c_split;
  // EPI = 1 (bias + relu epilogue): instead of C the kernel writes mask[m][n] = (acc + bias[n] > 0) and ONE partial
  // sum of max(acc + bias[n], 0) per block (partial[blockIdx.x]); see mdhip_matmul_bias_relu_sum
  const float *bias;
  uint8_t *mask;
  float *partial;
  float *sum_out;     // the 0-d result, written by the block that arrives last at `tickets` (md_ticket.h)
  unsigned *tickets;
  // diagnostic (MDHIP_GEMM_STAMP=1, never set in production): per block {shader cycles, 100 MHz ticks} around the whole kernel body
  unsigned long long *stamp;
  // ragged direct-to-LDS kernels: 16 B of zeros in device memory — what a DMA lane fetches for a position outside the operand
  const float *zero;
  // host side only: the row-contiguous operand A (B) sits in a buffer of OURS whose rows are padded to a multiple of four
  // elements (repacked copy, HipExec::gemm): a 16-B piece that straddles the M (N) edge reads the padding, which only feeds
  // output rows (columns) outside C — so M (N) need not be a multiple of 4 for the direct-to-LDS kernels
  int pad_m, pad_n;
  // C is addressed with unit stride along ROWS (a swapped "TT" product, HipExec::gemm) and everything is 16-B aligned: the direct-to-LDS
  // NN kernel stores the four consecutive rows a lane holds of one column as one vector
  int c_vec_rows;
};

// Tile loaders for a ROWS x BK operand tile, NT threads, 16 B per thread per pass.
// KC = the operand's k axis is the contiguous one in memory (row-major A, or B given as Bt).
template <int ROWS, int BK, int NT, bool KC, bool EDGE>
__device__ __forceinline__ void load_tile(const float *__restrict__ P, int64_t rs, int64_t ks, int64_t row0, int64_t k0,
                                          int64_t rows, int64_t K, f32x4 (&r)[ROWS * BK / (4 * NT)], bool vec_ok = true) {
  constexpr int PASSES = ROWS * BK / (4 * NT);
  constexpr int TPR = BK / 4;            // KC: threads per row
  constexpr int RPP = NT / TPR;          // KC: rows per pass
  constexpr int TPK = ROWS / 4;          // !KC: threads per k-row
  constexpr int KPP = NT / TPK;          // !KC: k-rows per pass
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < PASSES; ++i) {
    int row, k;
    if constexpr (KC) { row = t / TPR + RPP * i; k = (t % TPR) * 4; }
    else { k = t / TPK + KPP * i; row = (t % TPK) * 4; }
    if constexpr (!EDGE) {
      r[i] = *reinterpret_cast<const f32x4 *>(P + (row0 + row) * rs + (k0 + k) * ks);
    } else {
      // ragged M/N/K: a vector that lies wholly inside the operand is still ONE 16-B load (only the
      // last tile row/column and the K tail take the per-element path); unaligned operands always do
      const int64_t rr0 = row0 + row, kk0 = k0 + k;
      const bool inside = KC ? (rr0 < rows && kk0 + 3 < K) : (kk0 < K && rr0 + 3 < rows);
      if (vec_ok && inside) {
        r[i] = *reinterpret_cast<const f32x4 *>(P + rr0 * rs + kk0 * ks);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t rr = rr0 + (KC ? 0 : j), kk = kk0 + (KC ? j : 0);
          r[i][j] = (rr < rows && kk < K) ? P[rr * rs + kk * ks] : 0.0f;
        }
      }
    }
  }
}
// one pass (16 B per thread) of load_tile (whole aligned tiles only): lets the k-tile body place the global loads one by one
template <int ROWS, int BK, int NT, bool KC>
__device__ __forceinline__ void load_tile_pass(const float *__restrict__ P, int64_t rs, int64_t ks, int64_t row0, int64_t k0, f32x4 (&r)[ROWS * BK / (4 * NT)], int i) {
  constexpr int TPR = BK / 4, RPP = NT / TPR, TPK = ROWS / 4, KPP = NT / TPK;
  const int t = threadIdx.x;
  int row, k;
  if constexpr (KC) { row = t / TPR + RPP * i; k = (t % TPR) * 4; }
  else { k = t / TPK + KPP * i; row = (t % TPK) * 4; }
  r[i] = *reinterpret_cast<const f32x4 *>(P + (row0 + row) * rs + (k0 + k) * ks);
}
template <int ROWS, int BK, int NT, bool KC>
__device__ __forceinline__ void store_tile(float (*S)[ROWS + LDP], const f32x4 (&r)[ROWS * BK / (4 * NT)]) {
  constexpr int PASSES = ROWS * BK / (4 * NT);
  constexpr int TPR = BK / 4, RPP = NT / TPR, TPK = ROWS / 4, KPP = NT / TPK;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < PASSES; ++i) {
    if constexpr (KC) {
      const int row = t / TPR + RPP * i, k = (t % TPR) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) S[k + j][row] = r[i][j];
    } else {
      const int k = t / TPK + KPP * i, row = (t % TPK) * 4;
      *reinterpret_cast<f32x4 *>(&S[k][row]) = r[i];
    }
  }
}

// Fused epilogue of  loss = sum(where(X @ W + b > 0, X @ W + b, 0))  (reference call pattern: matmul
// definitions.py:487-492 -> add :424-427 -> greater :468-471 -> where :555-559 -> sum :403-407): the pre-activation
// never goes to memory; what the backward pass needs of it — the mask — does, as numpy.bool_ bytes, and the block's
// relu sum goes to partial[blockIdx.x]; the block that finishes last sums the partials in index order into sum_out.
// The accumulator layout gives a lane ONE column and 16 rows: written as it stands the mask would go out in 32-byte
// pieces of single bytes (measured: +106 us on the 8192 x 4096 x 4096 product). Instead the four lanes of a quad
// exchange their 16 result bits, each lane packs the 4 adjacent columns of 4 of the rows into one dword, the tile's
// mask is assembled in LDS (`sm`: an operand buffer, free by now; 16-byte groups XOR-swizzled by the row so that
// neither the dword writes nor the 16-byte reads conflict) and leaves as whole 128-byte rows. NPASS > 1: the tile's
// rows go through a scratch of BM / NPASS rows in NPASS rounds (256x256 tile: 64 KiB of mask, 32 KiB operand buffers).
// Called by every thread of the block, behind a barrier that retires the last reads of the operand buffers.
template <int BM, int BN, int WM, int WN, int NPASS>
__device__ __forceinline__ void md_epi_bias_relu(const f32x16 (&acc)[BM / (32 * WM)][BN / (32 * WN)], const GemmArgs &g, int64_t m0, int64_t n0,
                                                 uint32_t *sm, float *red) {
  constexpr int NT = 64 * WM * WN, WTM = BM / (32 * WM), WTN = BN / (32 * WN);
  constexpr int FP = WTM / NPASS;              // fragment rows of a wave per pass
  constexpr int SROWS = BM / NPASS;            // scratch rows
  static_assert(WTM % NPASS == 0 && BN % 16 == 0, "whole fragment rows per pass, 16-byte mask vectors");
  constexpr int RW = BN / 4;                   // dwords per mask row of the tile
  constexpr int GX = (BN / 16) < 8 ? (BN / 16) : 8;   // 16-byte groups per row that take part in the swizzle
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN, l32 = lane & 31, h = lane >> 5;
  const int q = l32 >> 2, jq = l32 & 3;
  float lsum = 0.0f;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    if (p > 0) __syncthreads();
#pragma unroll
    for (int ii = 0; ii < FP; ++ii) {
      const int i = p * FP + ii;
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        const int lcol = wn * (WTN * 32) + j * 32 + l32;
        const float bv = g.bias[n0 + lcol];
        uint32_t bits = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = acc[i][j][r] + bv;
          const bool on = z > 0.0f;
          lsum += on ? z : 0.0f;
          bits |= (uint32_t)on << r;
        }
        uint32_t qb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) qb[k] = (uint32_t)__shfl((int)bits, (lane & ~3) + k, 64);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = jq + 4 * t;                                   // this lane packs accumulator rows r = jq, jq+4, jq+8, jq+12
          const uint32_t d = ((qb[0] >> r) & 1u) | (((qb[1] >> r) & 1u) << 8) | (((qb[2] >> r) & 1u) << 16) | (((qb[3] >> r) & 1u) << 24);
          const int srow = wm * (FP * 32) + ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;   // scratch row
          const int w = (wn * (WTN * 32) + j * 32) / 4 + q;          // dword index inside the mask row
          const int grp = (w >> 2) ^ (srow & (GX - 1));               // swizzled 16-byte group
          sm[srow * RW + (grp << 2) + (w & 3)] = d;
        }
      }
    }
    __syncthreads();
    constexpr int VPR = BN / 16;  // 16-byte vectors per row
    for (int v = threadIdx.x; v < SROWS * VPR; v += NT) {
      const int srow = v / VPR, gv = v - srow * VPR;
      const int swm = srow / (FP * 32), rem = srow - swm * (FP * 32);
      const int64_t row = m0 + swm * (WTM * 32) + p * (FP * 32) + rem;
      const uint4 val = *reinterpret_cast<const uint4 *>(sm + srow * RW + ((gv ^ (srow & (GX - 1))) << 2));
      *reinterpret_cast<uint4 *>(g.mask + row * g.N + n0 + gv * 16) = val;
    }
  }
  // block sum in a fixed order: lanes by shuffle, then the waves through LDS
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) lsum += __shfl_down(lsum, d, 64);
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < WM * WN; ++w) t += red[w];
    md_st_sc1(g.partial + blockIdx.x, t);
  }
  // the block that arrives last sums the per-block partials in index order (md_ticket.h): no finishing launch
  unsigned *flag = reinterpret_cast<unsigned *>(red + 32);
  const unsigned nblk = gridDim.x;
  if (!(nblk >= 64 ? md_ticket_last2(g.tickets, blockIdx.x, nblk, flag) : md_ticket_last(g.tickets, nblk, flag))) return;
  float t = md_fold_partials<RSum>(g.partial, nblk);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d, 64);
  if (lane == 0) red[wave] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.0f;
#pragma unroll
    for (int w = 0; w < WM * WN; ++w) tot += red[w];
    g.sum_out[0] = tot;
  }
}

// BM x BN block tile, BK k-step, WM x WN waves, each wave (WTM*32) x (WTN*32).
// SPLITK is a separate instantiation on purpose: with the k-range arithmetic compiled into the plain
// kernel its main loop came out instruction-for-instruction the same but with another register
// assignment, and NN at 4096^3 dropped from 136.6 to 133.3 TFLOP/s (same-box A/B).
// (two waves per SIMD is what the schedule counts on for every tile up to 256 x 128 — 128 accumulator registers. The epilogue
// variants DECLARE it: left alone the bias/relu epilogue took 171 + 128 registers, one wave per SIMD, the whole product 5 % slower.
// The plain kernel keeps its allocation — 61 VGPRs + 128 AGPRs — untouched.)
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, bool EDGE, bool SPLITK = false, int SCHED = 0, int EPI = 0>
__global__ void __launch_bounds__(64 * WM * WN, (EPI != 0 && BM * BN <= 256 * 128 && WM * WN <= 4) ? 2 : 1) k_gemm_f32_mfma(GemmArgs g) {
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / (32 * WM), WTN = BN / (32 * WN);  // MFMA tiles per wave along m / n
  __shared__ float As[2][BK][BM + LDP];
  __shared__ float Bs[2][BK][BN + LDP];

  // XCD-aware tile order: ids b, b+8, b+16.. share an XCD -> give them neighbours
  const int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  int tm, tn;
  if (g.super_h > 1) {
    // tiles are numbered column-major inside bands of super_h tile rows, so that the nblk/8 consecutive
    // ids of one XCD form a compact block (8 x 8 tiles at 4096^2 instead of 2 x 32): fewer distinct
    // A/B panels per XCD L2, i.e. less L2 <- fabric traffic for the same work (speed-neutral: the
    // kernel is not fabric-bound)
    const int band = bid / (g.super_h * g.tiles_n), within = bid - band * (g.super_h * g.tiles_n);
    const int hgt = (band + 1) * g.super_h <= g.tiles_m ? g.super_h : g.tiles_m - band * g.super_h;
    tn = within / hgt;
    tm = band * g.super_h + within - tn * hgt;
  } else {
    tm = bid / g.tiles_n;
    tn = bid - tm * g.tiles_n;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t bz = blockIdx.z;
  const float *A = g.A + bz * g.a_bs;
  const float *B = g.B + bz * g.b_bs;
  float *C = g.C + bz * g.c_bs;
  int64_t Kl = g.K;  // this block's k extent
  if constexpr (SPLITK) {
    const int64_t ks0 = (int64_t)blockIdx.y * g.k_chunk;
    A += ks0 * g.a_ks;
    B += ks0 * g.b_ks;
    C += (int64_t)blockIdx.y * g.c_split;
    Kl = g.K - ks0 > g.k_chunk ? g.k_chunk : g.K - ks0;
  }

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l32 = lane & 31, h = lane >> 5;
  unsigned long long st_c = 0, st_r = 0;
  if (g.stamp) { st_c = __builtin_amdgcn_s_memtime(); st_r = __builtin_amdgcn_s_memrealtime(); }

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 ra[BM * BK / (4 * NT)], rb[BN * BK / (4 * NT)];
  const int64_t nk = (Kl + BK - 1) / BK;
  // A tile rows = m (row stride a_ms), B tile rows = n (row stride b_ns)
  // Pipeline: LDS buffer `cur` holds tile kt, registers hold tile kt+1 (landed), and
  // inside the MFMA stream of tile kt the wave (step 0/1) writes tile kt+1 to the
  // other LDS buffer and (step 2) issues the global loads of tile kt+2 - so staging
  // costs no MFMA time of its own and the loads have ~6 steps (>3000 cycles) to land.
  const bool vec_ok = g.vec_ok != 0;
  load_tile<BM, BK, NT, A_KC, EDGE>(A, g.a_ms, g.a_ks, m0, 0, g.M, Kl, ra, vec_ok);
  load_tile<BN, BK, NT, B_KC, EDGE>(B, g.b_ns, g.b_ks, n0, 0, g.N, Kl, rb, vec_ok);
  store_tile<BM, BK, NT, A_KC>(As[0], ra);
  store_tile<BN, BK, NT, B_KC>(Bs[0], rb);
  if (nk > 1) {
    load_tile<BM, BK, NT, A_KC, EDGE>(A, g.a_ms, g.a_ks, m0, BK, g.M, Kl, ra, vec_ok);
    load_tile<BN, BK, NT, B_KC, EDGE>(B, g.b_ns, g.b_ks, n0, BK, g.N, Kl, rb, vec_ok);
  }
  __syncthreads();

  int cur = 0;
  float fa[2][WTM], fb[2][WTN];
  if constexpr (SCHED != 0) {  // rotated loop: a tile's first fragments are read right behind the barrier that publishes it
#pragma unroll
    for (int i = 0; i < WTM; ++i) fa[0][i] = As[0][h][wm * (WTM * 32) + i * 32 + l32];
#pragma unroll
    for (int j = 0; j < WTN; ++j) fb[0][j] = Bs[0][h][wn * (WTN * 32) + j * 32 + l32];
  }
  for (int64_t kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
    if constexpr (SCHED == 0) {
#pragma unroll
      for (int i = 0; i < WTM; ++i) fa[0][i] = As[cur][h][wm * (WTM * 32) + i * 32 + l32];
#pragma unroll
      for (int j = 0; j < WTN; ++j) fb[0][j] = Bs[cur][h][wn * (WTN * 32) + j * 32 + l32];
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int c = (kk >> 1) & 1;
      if (kk + 2 < BK) {  // fragment reads one step ahead (register double buffer)
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[c ^ 1][i] = As[cur][kk + 2 + h][wm * (WTM * 32) + i * 32 + l32];
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[c ^ 1][j] = Bs[cur][kk + 2 + h][wn * (WTN * 32) + j * 32 + l32];
      }
      if constexpr (SCHED == 0) {
        if (kk == 0 && more) store_tile<BM, BK, NT, A_KC>(As[cur ^ 1], ra);
        if (kk == 2 && more) store_tile<BN, BK, NT, B_KC>(Bs[cur ^ 1], rb);
        if (kk == 4 && more2) {
          load_tile<BM, BK, NT, A_KC, EDGE>(A, g.a_ms, g.a_ks, m0, (kt + 2) * BK, g.M, Kl, ra, vec_ok);
          load_tile<BN, BK, NT, B_KC, EDGE>(B, g.b_ns, g.b_ks, n0, (kt + 2) * BK, g.N, Kl, rb, vec_ok);
        }
        // pin the order: staging and prefetch are issued ahead of this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // branch-free body (the staging of the last two tiles is redundant, never wrong: the other
        // LDS buffer is not read again, and the tile index is clamped) so that the whole k-tile is
        // ONE scheduling region, and every MFMA is followed by a slice of the step's memory work
        if (kk == 0) store_tile<BM, BK, NT, A_KC>(As[cur ^ 1], ra);
        if (kk == 2) store_tile<BN, BK, NT, B_KC>(Bs[cur ^ 1], rb);
        // small tiles (2-4 MFMAs per step): the refill of the staging registers starts the step after they were written to LDS,
        // one 16-B load per step — A's passes from step 1, then B's — instead of all of them in step 4: a load issued in step 4
        // has 4 steps x 2 MFMAs (~1000 cycles at two waves per SIMD) to land before its LDS store, which the memory latency
        // exceeds (ablation: the operand loads cost these tiles 11 %, the 256x128 tile 4 %); 2048^3: 112.3-113.1 -> 115.7-116.2
        // TFLOP/s. The 256x128 tile keeps the step-4 burst (spread: 139.6-140.2 -> 137.4-137.9).
        constexpr bool SPREAD = !EDGE && BM * BN <= 128 * 128;
        constexpr int PA = BM * BK / (4 * NT), PB = BN * BK / (4 * NT);
        if constexpr (SPREAD) {
          const int64_t ktl = more2 ? kt + 2 : nk - 1;
          const int sidx = kk >> 1;
          if (sidx >= 1 && sidx - 1 < PA) load_tile_pass<BM, BK, NT, A_KC>(A, g.a_ms, g.a_ks, m0, ktl * BK, ra, sidx - 1);
          if (sidx - 1 >= PA && sidx >= 2 && sidx - 1 - PA < PB) load_tile_pass<BN, BK, NT, B_KC>(B, g.b_ns, g.b_ks, n0, ktl * BK, rb, sidx - 1 - PA);
        } else if (kk == 4) {
          const int64_t ktl = more2 ? kt + 2 : nk - 1;
          load_tile<BM, BK, NT, A_KC, EDGE>(A, g.a_ms, g.a_ks, m0, ktl * BK, g.M, Kl, ra, vec_ok);
          load_tile<BN, BK, NT, B_KC, EDGE>(B, g.b_ns, g.b_ks, n0, ktl * BK, g.N, Kl, rb, vec_ok);
        }
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < WTM * WTN; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // one LDS read (next step's fragments)
          if (kk == 0 || kk == 2) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);  // LDS writes of the staged tile
          if (SPREAD ? (m == 0 && kk >= 2) : kk == 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // one global load of tile kt+2
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    if constexpr (SCHED != 0) {  // (the last step's MFMAs are register-only and sink below the barrier: they cover these reads)
#pragma unroll
      for (int i = 0; i < WTM; ++i) fa[0][i] = As[cur][h][wm * (WTM * 32) + i * 32 + l32];
#pragma unroll
      for (int j = 0; j < WTN; ++j) fb[0][j] = Bs[cur][h][wn * (WTN * 32) + j * 32 + l32];
    }
  }

  if (g.stamp && threadIdx.x == 0) {   // (the stamps go to a buffer of their own: MI355X_MICROARCH.md "DVFS give-back" item 6)
    g.stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c;
    g.stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r;
  }
  if constexpr (EPI == 1) {
    static_assert(sizeof(As) >= (size_t)BM * BN, "the tile's mask must fit the A staging buffers");
    __syncthreads();
    md_epi_bias_relu<BM, BN, WM, WN, 1>(acc, g, m0, n0, reinterpret_cast<uint32_t *>(&As[0][0][0]), &Bs[0][0][0]);
    return;
  }
  // C/D layout of the 32x32 accumulator: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int64_t col = n0 + wn * (WTN * 32) + j * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * (WTM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (!EDGE || (row < g.M && col < g.N)) C[row * g.c_ms + col * g.c_ns] = acc[i][j][r];
      }
    }
}

// ---- TN products (both operands row-contiguous along the tile's rows) staged global -> LDS directly ------------------
// global_load_lds_dwordx4: one wave-instruction moves 64 lanes x 16 B = 1 KiB of the unpadded [BK][ROWS] image (ROWS/256 of a
// k-row, or 256/ROWS whole k-rows): no staging registers, no ds_write. The LDS address is wave-uniform (M0) + lane * 16, the
// SOURCE address is per lane. One tile ahead: tile kt+1 lands in the other buffer while tile kt is multiplied, and the barrier
// that ends the k-tile waits for it (vmcnt(0), which __syncthreads() emits for an LDS-DMA in flight). The two buffers of an
// operand are SEPARATE __shared__ objects and the k-loop is unrolled by two with the buffer fixed at compile time: with one
// array indexed by a runtime `cur` the compiler cannot tell the fragment reads of buffer cur from the DMA writes to cur^1 and
// drains the DMA (s_waitcnt vmcnt(0)) in front of every step's ds_read (seen in the .s of the first version of this kernel).
typedef __attribute__((address_space(3))) void md_lds_void;
typedef __attribute__((address_space(1))) const void md_gbl_void;
// RAGGED (problem sizes that the tiles do not divide; operand rows a multiple of 4 so that no 16-B piece straddles the edge): a lane
// whose four rows or whose k lie outside the operand fetches 16 B of zeros instead (the SOURCE of a DMA is per lane) — the tile
// image is then complete, the main loop is the same, the epilogue drops the rows / columns outside C.
template <int ROWS, int BK, int NT, bool RAGGED = false>
__device__ __forceinline__ void glds_tile_pass(const float *__restrict__ P, int64_t ks, int64_t row0, int64_t k0, float *S, int i,
                                               int64_t rows_total = 0, int64_t k_total = 0, const float *zero = nullptr) {
  constexpr int NW = NT / 64;
  static_assert(ROWS * BK % (256 * NW) == 0 && (ROWS & (ROWS - 1)) == 0, "whole 1-KiB pieces per wave");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int piece = i * NW + wave;
  const int f = piece * 256 + lane * 4, k = f / ROWS, r = f % ROWS;
  const float *src = P + (row0 + r) + (k0 + k) * ks;
  if constexpr (RAGGED) src = (row0 + r < rows_total && k0 + k < k_total) ? src : zero;
  __builtin_amdgcn_global_load_lds((md_gbl_void *)src, (md_lds_void *)(S + piece * 256), 16, 0, 0);
}

// The same piece with the address split into a WAVE-UNIFORM base (scalar registers; advanced per pass and per k-tile by scalar
// arithmetic) and a loop-invariant 32-bit per-lane byte offset: the DMA then takes its address as `v_offset, s[base]` and costs no
// vector ALU work. With the full 64-bit address per lane (above) the compiler re-derived it for every piece — a 32-bit multiply, a
// 64-bit multiply-add and two 64-bit shift-adds, ~45 cycles of VALU per DMA, which a lone wave per SIMD cannot hide behind a 64-cycle
// MFMA: ablation (profiles/r2_gemm_glds_ab.log, fifth part) priced the DMAs at 3.9 % of the 256x256 kernel and 8.7 % of the eight-wave
// 128x128 one. Whole-tile kernels only (the ragged variants select between two addresses per lane).
template <int ROWS> __device__ __forceinline__ uint32_t glds_tile_lane_off(int64_t ks) {
  const int lane = threadIdx.x & 63;
  constexpr int LPR = ROWS / 4 > 64 ? 64 : ROWS / 4;   // lanes per k-row
  return (uint32_t)(((int64_t)(lane / LPR) * ks + (lane % LPR) * 4) * 4);
}
// A wave-uniform 64-bit address, made opaque to the optimiser as two scalar words: the DMA's address is then visibly
// (scalar base) + (zero-extended 32-bit lane offset) and nothing else — left transparent, the compiler re-associated the sum of
// several pieces as ((lane offset + k offset) + piece base), a 64-bit VECTOR add per piece, and the DMA took the per-lane 64-bit
// address form (half the pieces of the NT kernel, scripts/isa_check.py). No instruction is emitted for this.
__device__ __forceinline__ const char *md_opaque_uniform(const char *p) {
  uint32_t lo = (uint32_t)(uintptr_t)p, hi = (uint32_t)((uintptr_t)p >> 32);
  asm("" : "+s"(lo), "+s"(hi));
  return reinterpret_cast<const char *>(((uintptr_t)hi << 32) | lo);
}
// `wbase` = the wave's first piece of tile 0 (uniform pointer: P + row0 + wave * KPP * ks), `step` = floats between a wave's consecutive
// pieces (NW * KPP * ks), `koff` = floats from tile 0 to this k-tile (k0 * ks): adds only, in scalar registers.
template <int ROWS, int BK, int NT, bool PRED = false>
__device__ __forceinline__ void glds_tile_pass_u(const float *wbase, int64_t step, int64_t koff, float *S, int i, uint32_t lane_off, bool pred = true) {
  constexpr int NW = NT / 64;
  static_assert(ROWS <= 256, "a piece covers whole k-rows");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char *ub = md_opaque_uniform(reinterpret_cast<const char *>(wbase + koff + (int64_t)i * step));
  // keeps the 32 -> 64-bit extension of the lane offset next to the DMA (address mode `v_offset, s[base]`; hoisted out of the loop it
  // arrives as a 64-bit register pair and the DMA takes the slow per-lane 64-bit form): an empty, NON-volatile statement that "depends" on
  // the k-tile offset, so it can neither leave the loop nor pin the instruction order
  asm("" : "+v"(lane_off) : "s"((int)koff));
  // `pred` (rows-only ragged kernels): a lane whose four rows lie outside the operand fetches nothing — its LDS slot keeps whatever an
  // earlier tile left there, which only reaches outputs outside C (row m of A feeds row m of C alone, column n of B column n alone)
  if constexpr (PRED) {
    if (pred) __builtin_amdgcn_global_load_lds((md_gbl_void *)(ub + lane_off), (md_lds_void *)(S + (i * NW + wave) * 256), 16, 0, 0);
  } else {
    __builtin_amdgcn_global_load_lds((md_gbl_void *)(ub + lane_off), (md_lds_void *)(S + (i * NW + wave) * 256), 16, 0, 0);
  }
}

template <int V> struct MdInt { static constexpr int value = V; };

// RAGGED: 0 whole tiles | 1 any ragged size (zero-filled lanes, 64-bit lane addresses) | 2 ragged M / N with K % BK == 0 (predicated lanes,
// scalar-base addresses: the fast form)
// (NBUF = 3: three LDS buffers, DMA two k-tiles ahead, counted vmcnt at the boundary — see k_gemm_f32_kc_glds)
template <int BM, int BN, int BK, int WM, int WN, int RAGGED = 0, int NBUF = 2>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_f32_tn_glds(GemmArgs g) {
  static_assert(NBUF == 2 || (NBUF == 3 && RAGGED == 0), "three buffers: whole-tile kernel only");
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / (32 * WM), WTN = BN / (32 * WN);
  constexpr int PA = BM * BK / (4 * NT), PB = BN * BK / (4 * NT);
  // the next tile's DMA pieces go out within the first two steps of the k-tile (with the scalar-base addressing a DMA costs next to
  // nothing to issue; the eight-wave tile gained 7 % at 2048^3 over spreading them, the 256x256 tile is indifferent)
  constexpr int DMA_STEPS = 2;
  constexpr int PPS = (PA + PB + DMA_STEPS - 1) / DMA_STEPS;
  __shared__ __attribute__((aligned(16))) float A0[BK][BM];   // (DMA destinations: 16-B pieces)
  __shared__ __attribute__((aligned(16))) float A1[BK][BM];
  __shared__ __attribute__((aligned(16))) float B0[BK][BN];
  __shared__ __attribute__((aligned(16))) float B1[BK][BN];
  __shared__ __attribute__((aligned(16))) float A2[NBUF == 3 ? BK : 1][BM];
  __shared__ __attribute__((aligned(16))) float B2[NBUF == 3 ? BK : 1][BN];

  const int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // XCD-aware tile order, as in k_gemm_f32_mfma
  int tm, tn;
  if (g.super_h > 1) {
    const int band = bid / (g.super_h * g.tiles_n), within = bid - band * (g.super_h * g.tiles_n);
    const int hgt = (band + 1) * g.super_h <= g.tiles_m ? g.super_h : g.tiles_m - band * g.super_h;
    tn = within / hgt;
    tm = band * g.super_h + within - tn * hgt;
  } else {
    tm = bid / g.tiles_n;
    tn = bid - tm * g.tiles_n;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t bz = blockIdx.z;
  const float *A = g.A + bz * g.a_bs;
  const float *B = g.B + bz * g.b_bs;
  float *C = g.C + bz * g.c_bs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l32 = lane & 31, h = lane >> 5;
  const int am = wm * (WTM * 32) + l32, bn = wn * (WTN * 32) + l32;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int64_t nk = RAGGED == 1 ? (g.K + BK - 1) / BK : g.K / BK;   // (forms 0 and 2: whole k-tiles only, the launcher checks)
#pragma unroll
  for (int i = 0; i < PA; ++i) glds_tile_pass<BM, BK, NT, RAGGED != 0>(A, g.a_ks, m0, 0, &A0[0][0], i, g.M, g.K, g.zero);
#pragma unroll
  for (int i = 0; i < PB; ++i) glds_tile_pass<BN, BK, NT, RAGGED != 0>(B, g.b_ks, n0, 0, &B0[0][0], i, g.N, g.K, g.zero);
  if constexpr (NBUF == 3) {   // tile 1 into buffer 1 (K >= 2 k-tiles: the launcher checks)
#pragma unroll
    for (int i = 0; i < PA; ++i) glds_tile_pass<BM, BK, NT, false>(A, g.a_ks, m0, BK, &A1[0][0], i, g.M, g.K, g.zero);
#pragma unroll
    for (int i = 0; i < PB; ++i) glds_tile_pass<BN, BK, NT, false>(B, g.b_ks, n0, BK, &B1[0][0], i, g.N, g.K, g.zero);
  }
  const uint32_t la = glds_tile_lane_off<BM>(g.a_ks), lb = glds_tile_lane_off<BN>(g.b_ks);   // per-lane parts of the DMA addresses (loop-invariant)
  const int uw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float *wa = A + m0 + (int64_t)uw * (256 / BM) * g.a_ks, *wb = B + n0 + (int64_t)uw * (256 / BN) * g.b_ks;   // uniform parts
  const int64_t sa = (int64_t)(NT / 64) * (256 / BM) * g.a_ks, sb = (int64_t)(NT / 64) * (256 / BN) * g.b_ks;
  // rows-only ragged form: which lanes' four rows lie inside the operands (the same lanes in every piece of a [k][row] image)
  bool pra = true, prb = true;
  if constexpr (RAGGED == 2) {
    pra = m0 + (int64_t)((threadIdx.x & 63) % (BM / 4 > 64 ? 64 : BM / 4)) * 4 < g.M;
    prb = n0 + (int64_t)((threadIdx.x & 63) % (BN / 4 > 64 ? 64 : BN / 4)) * 4 < g.N;
  }
  if constexpr (NBUF == 3) {
    constexpr int NW8 = PA + PB;
    __builtin_amdgcn_s_waitcnt((NW8 & 0xF) | ((NW8 >> 4) << 14) | (7 << 4) | (15 << 8));   // tile 0 landed; tile 1 may still fly
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();
  }

  // fragments for TWO steps (4 k) per LDS instruction: rows 4p+h and 4p+2+h of the [k][row] image lie 2*ROWS floats apart, a
  // multiple of 64 dwords, so the two loads of a lane fuse into one ds_read2st64_b32 — half the LDS read instructions; MFMA t of
  // the pair takes k = 4p+2t (lanes 0-31) and 4p+2t+1 (lanes 32-63): the plain k order
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int NPR = BK / 4;
  f32x2 fa[2][WTM], fb[2][WTN];
#define MD_TN_READ(BUF, p, c)                                                                                          \
  {                                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) {                                                                  \
      fa[c][i][0] = ((BUF) == 0 ? A0 : (BUF) == 1 ? A1 : A2)[4 * (p) + h][am + i * 32];                                                       \
      fa[c][i][1] = ((BUF) == 0 ? A0 : (BUF) == 1 ? A1 : A2)[4 * (p) + 2 + h][am + i * 32];                                                   \
    }                                                                                                                  \
    _Pragma("unroll") for (int q = 0; q < WTN; ++q) {                                                                  \
      fb[c][q][0] = ((BUF) == 0 ? B0 : (BUF) == 1 ? B1 : B2)[4 * (p) + h][bn + q * 32];                                                       \
      fb[c][q][1] = ((BUF) == 0 ? B0 : (BUF) == 1 ? B1 : B2)[4 * (p) + 2 + h][bn + q * 32];                                                   \
    }                                                                                                                  \
  }
  MD_TN_READ(0, 0, 0)
  auto ktile = [&](auto curc, int64_t kn) {
    constexpr int CUR = decltype(curc)::value;
#pragma unroll
    for (int p = 0; p < NPR; ++p) {
      const int c = p & 1;
      if (p + 1 < NPR) MD_TN_READ(CUR, p + 1, c ^ 1)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int sidx = 2 * p + t;
        int n_dma = 0;
#pragma unroll
        for (int q = 0; q < PPS; ++q) {
          const int pi = sidx * PPS + q;
          if (pi < PA) {
            if constexpr (RAGGED == 1) glds_tile_pass<BM, BK, NT, true>(A, g.a_ks, m0, kn * BK, (NBUF == 2 ? (CUR ? &A0[0][0] : &A1[0][0]) : (CUR == 0 ? &A2[0][0] : CUR == 1 ? &A0[0][0] : &A1[0][0])), pi, g.M, g.K, g.zero);
            else if constexpr (RAGGED == 2) glds_tile_pass_u<BM, BK, NT, true>(wa, sa, kn * BK * g.a_ks, (NBUF == 2 ? (CUR ? &A0[0][0] : &A1[0][0]) : (CUR == 0 ? &A2[0][0] : CUR == 1 ? &A0[0][0] : &A1[0][0])), pi, la, pra);
            else glds_tile_pass_u<BM, BK, NT>(wa, sa, kn * BK * g.a_ks, (NBUF == 2 ? (CUR ? &A0[0][0] : &A1[0][0]) : (CUR == 0 ? &A2[0][0] : CUR == 1 ? &A0[0][0] : &A1[0][0])), pi, la);
            ++n_dma;
          } else if (pi < PA + PB) {
            if constexpr (RAGGED == 1) glds_tile_pass<BN, BK, NT, true>(B, g.b_ks, n0, kn * BK, (NBUF == 2 ? (CUR ? &B0[0][0] : &B1[0][0]) : (CUR == 0 ? &B2[0][0] : CUR == 1 ? &B0[0][0] : &B1[0][0])), pi - PA, g.N, g.K, g.zero);
            else if constexpr (RAGGED == 2) glds_tile_pass_u<BN, BK, NT, true>(wb, sb, kn * BK * g.b_ks, (NBUF == 2 ? (CUR ? &B0[0][0] : &B1[0][0]) : (CUR == 0 ? &B2[0][0] : CUR == 1 ? &B0[0][0] : &B1[0][0])), pi - PA, lb, prb);
            else glds_tile_pass_u<BN, BK, NT>(wb, sb, kn * BK * g.b_ks, (NBUF == 2 ? (CUR ? &B0[0][0] : &B1[0][0]) : (CUR == 0 ? &B2[0][0] : CUR == 1 ? &B0[0][0] : &B1[0][0])), pi - PA, lb);
            ++n_dma;
          }
        }
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][t], fb[c][j][t], acc[i][j], 0, 0, 0);
        constexpr int NRD = WTM + WTN, MPS = WTM * WTN;
        const int rd_slots = MPS >= 8 ? MPS : p + 1 < NPR ? (NRD - t * MPS < 0 ? 0 : (NRD - t * MPS > MPS ? MPS : NRD - t * MPS)) : 0;
#pragma unroll
        for (int m = 0; m < MPS; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (m < rd_slots) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (m < n_dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
    }
    if constexpr (NBUF == 2) {
      __syncthreads();
      MD_TN_READ(CUR ^ 1, 0, 0)
    } else {
      constexpr int NW8 = PA + PB;   // (counted wait: the tile computed next is complete, the one just issued may still fly)
      __builtin_amdgcn_s_waitcnt((NW8 & 0xF) | ((NW8 >> 4) << 14) | (7 << 4) | (15 << 8));
      __builtin_amdgcn_s_barrier();
      MD_TN_READ((CUR + 1) % 3, 0, 0)
    }
  };
  int64_t kt = 0;
  if constexpr (NBUF == 2) {
    for (; kt + 1 < nk; kt += 2) {
      ktile(MdInt<0>{}, kt + 1);
      ktile(MdInt<1>{}, kt + 2 < nk ? kt + 2 : nk - 1);
    }
    if (kt < nk) ktile(MdInt<0>{}, nk - 1);
  } else {
    for (; kt + 2 < nk; kt += 3) {
      ktile(MdInt<0>{}, kt + 2);
      ktile(MdInt<1>{}, kt + 3 < nk ? kt + 3 : nk - 1);
      ktile(MdInt<2>{}, kt + 4 < nk ? kt + 4 : nk - 1);
    }
    if (kt < nk) ktile(MdInt<0>{}, nk - 1);
    if (kt + 1 < nk) ktile(MdInt<1>{}, nk - 1);
  }
#undef MD_TN_READ

#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int64_t col = n0 + wn * (WTN * 32) + j * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * (WTM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (RAGGED == 0 || (row < g.M && col < g.N)) C[row * g.c_ms + col * g.c_ns] = acc[i][j][r];
      }
    }
}

// ---- NN / NT products: the k-contiguous operand(s) direct to LDS as well -------------------------------------------------
// A lane-linear DMA cannot transpose, so the k-contiguous operand keeps its memory order in LDS: 1-KiB pieces of 16 rows x 16 k,
// stored [k/4][row][4] (lane = (k/4) * 16 + row: the source of one wave-instruction is 16 rows x 64 B, as with register staging).
// A fragment is then ONE ds_read_b128 per 32 rows and EIGHT k: the lanes of half h take the four k of group 2j+h, and MFMA t of
// the pair multiplies k = 8j + t (lanes 0-31) and k = 8j + 4 + t (lanes 32-63) — v_mfma_f32_32x32x2 does not care WHICH two k a
// call carries, only that A and B agree. The sum over k therefore runs in the order (0,4),(1,5),(2,6),(3,7),(8,12).. instead of
// (0,1),(2,3)..: a different rounding sequence from the register-staged kernel (both are plain f32 fma chains; integers stay exact).
// B is either k-contiguous too (NT: same image) or row-contiguous (NN: the [k][n] image of k_gemm_f32_tn_glds, b32 reads at
// k = 8j + 4h + t). 16 consecutive lanes of a b128 read cover one 256-B run of a piece: no bank conflicts.
template <int ROWS, int BK, int NT, bool RAGGED = false>
__device__ __forceinline__ void glds_kc_pass(const float *__restrict__ P, int64_t rs, int64_t row0, int64_t k0, float *S, int i,
                                             int64_t rows_total = 0, int64_t k_total = 0, const float *zero = nullptr) {
  constexpr int NW = NT / 64, KH = BK / 16;
  static_assert(ROWS * BK % (256 * NW) == 0 && BK % 16 == 0, "whole 1-KiB pieces per wave");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  // a wave takes BOTH k-halves of a 16-row block back to back (the two pieces share their 128-B lines)
  const int rb = (i / KH) * NW + wave, kh = i % KH;
  const int row = rb * 16 + (lane & 15), k = kh * 16 + (lane >> 4) * 4;
  const float *src = P + (row0 + row) * rs + k0 + k;
  if constexpr (RAGGED) src = (row0 + row < rows_total && k0 + k < k_total) ? src : zero;   // (K % 4 == 0: no piece straddles the end)
  __builtin_amdgcn_global_load_lds((md_gbl_void *)src, (md_lds_void *)(S + (rb * KH + kh) * 256), 16, 0, 0);
}

__device__ __forceinline__ uint32_t glds_kc_lane_off(int64_t rs) {
  const int lane = threadIdx.x & 63;
  return (uint32_t)(((int64_t)(lane & 15) * rs + (lane >> 4) * 4) * 4);
}
// `wbase` = P + (row0 + wave * 16) * rs (uniform), `step` = NW * 16 * rs floats between a wave's 16-row blocks, `k0` = first k of the tile
template <int ROWS, int BK, int NT, bool PRED = false>
__device__ __forceinline__ void glds_kc_pass_u(const float *wbase, int64_t step, int64_t k0, float *S, int i, uint32_t lane_off, int rows_left = 0) {
  constexpr int NW = NT / 64, KH = BK / 16;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rb = (i / KH) * NW + wave, kh = i % KH;
  // (the k-half's 64 bytes ride in the LANE offset: with them in the base, two pieces 64 B apart were folded into one 64-bit per-lane
  // address plus a constant and the DMAs fell back to the `v[N:N+1], off` form with a v_lshl_add_u64 each: scripts/isa_check.py)
  const char *ub = md_opaque_uniform(reinterpret_cast<const char *>(wbase + (int64_t)(i / KH) * step + k0));
  lane_off += (uint32_t)(kh * 64);
  asm("" : "+v"(lane_off) : "s"((int)k0));   // (as in glds_tile_pass_u)
  // `rows_left` = rows of the operand from this tile's first row on (rows-only ragged kernels; see glds_tile_pass_u)
  if (!PRED || (int)(threadIdx.x & 15) + rb * 16 < rows_left) __builtin_amdgcn_global_load_lds((md_gbl_void *)(ub + lane_off), (md_lds_void *)(S + (rb * KH + kh) * 256), 16, 0, 0);
}

// NBUF = 3 (whole tiles, no epilogue; MDHIP_GEMM_NBUF=3, experiment): THREE LDS buffers per operand, the DMA runs TWO k-tiles ahead and the
// k-tile boundary waits with a COUNTED `s_waitcnt vmcnt(PA + PB)` — the tile needed next is complete, the one just issued may still be
// in flight — instead of the full drain a `__syncthreads()` brings (SQ counters: the small tiles park 20 % of their cycles there).
// CT: C is addressed with unit stride along rows (a "TT" product run as the swapped NN product, HipExec::gemm): 16-B stores of the
// four consecutive rows a lane holds. A separate instantiation — as a run-time branch it cost the 256x256 NN kernel its last registers.
template <int BM, int BN, int BK, int WM, int WN, bool B_KC, int EPI = 0, int RAGGED = 0, int NBUF = 2, bool CT = false>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_f32_kc_glds(GemmArgs g) {
  static_assert(NBUF == 2 || (NBUF == 3 && EPI == 0 && RAGGED == 0), "three buffers: plain whole-tile kernel only");
  static_assert(!CT || (!B_KC && EPI == 0 && RAGGED == 0), "transposed C addressing: whole-tile NN form only");
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / (32 * WM), WTN = BN / (32 * WN);
  constexpr int PA = BM * BK / (4 * NT), PB = BN * BK / (4 * NT), NP = BK / 8, KH = BK / 16;
  // the next tile's DMA pieces go out within the first DMA_STEPS steps of the k-tile (16 steps at BK 32); same-box A/B at 4096^3 NT,
  // r2_gemm_glds_ab.log: 16 steps 138.9 | 8: 144.6 | 4: 146.0 | 3: 145.5 | 2: 143.8 | 1: 140.4 TFLOP/s
  constexpr int DMA_STEPS = 4;   // (re-checked after the scalar-base addressing: 2 -> NT 146.7, 8 -> 147.9, 4 -> 149.2; all of the
                                 //  eight-wave tile's four pieces behind ONE step: NN 124 -> 114, NT 121 -> 99 at 2048^3)
  constexpr int PPS = (PA + PB + DMA_STEPS - 1) / DMA_STEPS;   // DMA pieces per step
  static_assert(NP % 2 == 0, "an even number of k-pairs per tile (fragment double buffer)");
  __shared__ __attribute__((aligned(16))) float A0[BM * BK];
  __shared__ __attribute__((aligned(16))) float A1[BM * BK];
  __shared__ __attribute__((aligned(16))) float B0[BN * BK];
  __shared__ __attribute__((aligned(16))) float B1[BN * BK];
  __shared__ __attribute__((aligned(16))) float A2[NBUF == 3 ? BM * BK : 4];
  __shared__ __attribute__((aligned(16))) float B2[NBUF == 3 ? BN * BK : 4];

  const int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  int tm, tn;
  if (g.super_h > 1) {
    const int band = bid / (g.super_h * g.tiles_n), within = bid - band * (g.super_h * g.tiles_n);
    const int hgt = (band + 1) * g.super_h <= g.tiles_m ? g.super_h : g.tiles_m - band * g.super_h;
    tn = within / hgt;
    tm = band * g.super_h + within - tn * hgt;
  } else {
    tm = bid / g.tiles_n;
    tn = bid - tm * g.tiles_n;
  }
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t bz = blockIdx.z;
  const float *A = g.A + bz * g.a_bs;
  const float *B = g.B + bz * g.b_bs;
  float *C = g.C + bz * g.c_bs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l32 = lane & 31, h = lane >> 5;
  // lane bases (floats) into the images; fragment i / pair j add compile-time offsets
  const int arow = wm * (WTM * 32) + l32;
  const int abase = (arow >> 4) * (KH * 256) + (arow & 15) * 4 + h * 64;
  const int brow = wn * (WTN * 32) + l32;
  const int bbase = B_KC ? (brow >> 4) * (KH * 256) + (brow & 15) * 4 + h * 64 : h * 4 * BN + brow;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int64_t nk = RAGGED == 1 ? (g.K + BK - 1) / BK : g.K / BK;
  unsigned long long st_c = 0, st_r = 0;   // diagnostic stamps (MDHIP_GEMM_STAMP=1), as in k_gemm_f32_mfma
  if (g.stamp) { st_c = __builtin_amdgcn_s_memtime(); st_r = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll
  for (int i = 0; i < PA; ++i) glds_kc_pass<BM, BK, NT, RAGGED != 0>(A, g.a_ms, m0, 0, A0, i, g.M, g.K, g.zero);
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    if constexpr (B_KC) glds_kc_pass<BN, BK, NT, RAGGED != 0>(B, g.b_ns, n0, 0, B0, i, g.N, g.K, g.zero);
    else glds_tile_pass<BN, BK, NT, RAGGED != 0>(B, g.b_ks, n0, 0, B0, i, g.N, g.K, g.zero);
  }
  if constexpr (NBUF == 3) {   // tile 1 into buffer 1 (K >= 2 k-tiles: the launcher checks)
#pragma unroll
    for (int i = 0; i < PA; ++i) glds_kc_pass<BM, BK, NT, false>(A, g.a_ms, m0, BK, A1, i, g.M, g.K, g.zero);
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if constexpr (B_KC) glds_kc_pass<BN, BK, NT, false>(B, g.b_ns, n0, BK, B1, i, g.N, g.K, g.zero);
      else glds_tile_pass<BN, BK, NT, false>(B, g.b_ks, n0, BK, B1, i, g.N, g.K, g.zero);
    }
  }
  const uint32_t la = glds_kc_lane_off(g.a_ms), lb = B_KC ? glds_kc_lane_off(g.b_ns) : glds_tile_lane_off<BN>(g.b_ks);   // (loop-invariant lane parts of the DMA addresses)
  const int uw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float *wa = A + (m0 + (int64_t)uw * 16) * g.a_ms;   // uniform parts: adds only inside the loop
  const float *wb = B_KC ? B + (n0 + (int64_t)uw * 16) * g.b_ns : B + n0 + (int64_t)uw * (256 / BN) * g.b_ks;
  const int64_t sa = (int64_t)(NT / 64) * 16 * g.a_ms, sb = B_KC ? (int64_t)(NT / 64) * 16 * g.b_ns : (int64_t)(NT / 64) * (256 / BN) * g.b_ks;
  // rows-only ragged form (RAGGED == 2): rows of the operands from this tile's first row on / lanes of the [k][n] image inside B
  int rla = 0, rlb = 0;
  bool prb = true;
  if constexpr (RAGGED == 2) {
    rla = (int)(g.M - m0 < (1 << 30) ? g.M - m0 : (1 << 30));
    rlb = (int)(g.N - n0 < (1 << 30) ? g.N - n0 : (1 << 30));
    if constexpr (!B_KC) prb = n0 + (int64_t)((threadIdx.x & 63) % (BN / 4 > 64 ? 64 : BN / 4)) * 4 < g.N;
  }
  if constexpr (NBUF == 3) {
    constexpr int NW8 = PA + PB;
    __builtin_amdgcn_s_waitcnt((NW8 & 0xF) | ((NW8 >> 4) << 14) | (7 << 4) | (15 << 8));   // tile 0 landed; tile 1 may still fly
    __builtin_amdgcn_s_barrier();
  } else {
    __syncthreads();
  }

  f32x4 fa[2][WTM], fb[2][WTN];
  // fragments of k-pair j out of buffer BUF (compile-time) into register set c
#define MD_KC_READ(BUF, j, c)                                                                                                   \
  {                                                                                                                             \
    const int joff = ((j) >> 1) * 256 + (((2 * (j)) & 3) * 64);   /* (a constant once the pair loop is unrolled) */             \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i)                                                                             \
      fa[c][i] = *reinterpret_cast<const f32x4 *>(((BUF) == 0 ? A0 : (BUF) == 1 ? A1 : A2) + abase + i * (2 * KH * 256) + joff);                       \
    _Pragma("unroll") for (int q = 0; q < WTN; ++q) {                                                                           \
      if constexpr (B_KC) fb[c][q] = *reinterpret_cast<const f32x4 *>(((BUF) == 0 ? B0 : (BUF) == 1 ? B1 : B2) + bbase + q * (2 * KH * 256) + joff);   \
      else {                                                                                                                    \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) fb[c][q][t] = ((BUF) == 0 ? B0 : (BUF) == 1 ? B1 : B2)[bbase + (8 * (j) + t) * BN + q * 32];     \
      }                                                                                                                         \
    }                                                                                                                           \
  }
  MD_KC_READ(0, 0, 0)

  auto ktile = [&](auto curc, int64_t kn) {
    constexpr int CUR = decltype(curc)::value;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int c = j & 1;
      if (j + 1 < NP) MD_KC_READ(CUR, j + 1, c ^ 1)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int sidx = j * 4 + t;
        int n_dma = 0;
#pragma unroll
        for (int q = 0; q < PPS; ++q) {
          const int pi = sidx * PPS + q;
          if (pi < PA) {
            if constexpr (RAGGED == 1) glds_kc_pass<BM, BK, NT, true>(A, g.a_ms, m0, kn * BK, (NBUF == 2 ? (CUR ? A0 : A1) : (CUR == 0 ? A2 : CUR == 1 ? A0 : A1)), pi, g.M, g.K, g.zero);
            else if constexpr (RAGGED == 2) glds_kc_pass_u<BM, BK, NT, true>(wa, sa, kn * BK, (NBUF == 2 ? (CUR ? A0 : A1) : (CUR == 0 ? A2 : CUR == 1 ? A0 : A1)), pi, la, rla);
            else glds_kc_pass_u<BM, BK, NT>(wa, sa, kn * BK, (NBUF == 2 ? (CUR ? A0 : A1) : (CUR == 0 ? A2 : CUR == 1 ? A0 : A1)), pi, la);
            ++n_dma;
          } else if (pi < PA + PB) {
            if constexpr (RAGGED == 1) {
              if constexpr (B_KC) glds_kc_pass<BN, BK, NT, true>(B, g.b_ns, n0, kn * BK, (NBUF == 2 ? (CUR ? B0 : B1) : (CUR == 0 ? B2 : CUR == 1 ? B0 : B1)), pi - PA, g.N, g.K, g.zero);
              else glds_tile_pass<BN, BK, NT, true>(B, g.b_ks, n0, kn * BK, (NBUF == 2 ? (CUR ? B0 : B1) : (CUR == 0 ? B2 : CUR == 1 ? B0 : B1)), pi - PA, g.N, g.K, g.zero);
            } else if constexpr (RAGGED == 2) {
              if constexpr (B_KC) glds_kc_pass_u<BN, BK, NT, true>(wb, sb, kn * BK, (NBUF == 2 ? (CUR ? B0 : B1) : (CUR == 0 ? B2 : CUR == 1 ? B0 : B1)), pi - PA, lb, rlb);
              else glds_tile_pass_u<BN, BK, NT, true>(wb, sb, kn * BK * g.b_ks, (NBUF == 2 ? (CUR ? B0 : B1) : (CUR == 0 ? B2 : CUR == 1 ? B0 : B1)), pi - PA, lb, prb);
            } else {
              if constexpr (B_KC) glds_kc_pass_u<BN, BK, NT>(wb, sb, kn * BK, (NBUF == 2 ? (CUR ? B0 : B1) : (CUR == 0 ? B2 : CUR == 1 ? B0 : B1)), pi - PA, lb);
              else glds_tile_pass_u<BN, BK, NT>(wb, sb, kn * BK * g.b_ks, (NBUF == 2 ? (CUR ? B0 : B1) : (CUR == 0 ? B2 : CUR == 1 ? B0 : B1)), pi - PA, lb);
            }
            ++n_dma;
          }
        }
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int q = 0; q < WTN; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][t], fb[c][q][t], acc[i][q], 0, 0, 0);
        // the next pair's fragment reads go behind the FIRST MFMAs of this pair, one each (with a read slot behind every MFMA
        // of the pair the scheduler parked the reads at its end, right in front of the wait for them)
        constexpr int NRD = WTM + (B_KC ? WTN : 4 * WTN), MPS = WTM * WTN;
        // tiles with >= 8 MFMAs per step: a read slot behind EVERY MFMA (measured best there); smaller ones: only behind the
        // first MFMAs of the pair, one per read
        constexpr bool RD_EVERY = MPS >= 8;
        const int rd_slots = RD_EVERY ? MPS : j + 1 < NP ? (NRD - t * MPS < 0 ? 0 : (NRD - t * MPS > MPS ? MPS : NRD - t * MPS)) : 0;
#pragma unroll
        for (int m = 0; m < MPS; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (m < rd_slots) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (m < n_dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
      }
    }
    if constexpr (NBUF == 2) {
      __syncthreads();
      MD_KC_READ(CUR ^ 1, 0, 0)
    } else {
      // counted wait: this wave's DMAs of the tile computed NEXT are done (issued a whole k-tile ago), the PA + PB it issued in this
      // tile may still fly; the barrier then publishes every wave's share of the next tile and frees this tile's buffer
      constexpr int NW8 = PA + PB;
      __builtin_amdgcn_s_waitcnt((NW8 & 0xF) | ((NW8 >> 4) << 14) | (7 << 4) | (15 << 8));
      __builtin_amdgcn_s_barrier();
      MD_KC_READ((CUR + 1) % 3, 0, 0)
    }
  };
  int64_t kt = 0;
  if constexpr (NBUF == 2) {
    for (; kt + 1 < nk; kt += 2) {
      ktile(MdInt<0>{}, kt + 1);
      ktile(MdInt<1>{}, kt + 2 < nk ? kt + 2 : nk - 1);
    }
    if (kt < nk) ktile(MdInt<0>{}, nk - 1);
  } else {
    // tile T is computed out of buffer T % 3 while the DMAs of tile T + 2 go out (past the end: a discarded re-read of the last tile)
    for (; kt + 2 < nk; kt += 3) {
      ktile(MdInt<0>{}, kt + 2);
      ktile(MdInt<1>{}, kt + 3 < nk ? kt + 3 : nk - 1);
      ktile(MdInt<2>{}, kt + 4 < nk ? kt + 4 : nk - 1);
    }
    if (kt < nk) ktile(MdInt<0>{}, nk - 1);
    if (kt + 1 < nk) ktile(MdInt<1>{}, nk - 1);
  }
#undef MD_KC_READ

  if (g.stamp && threadIdx.x == 0) {
    g.stamp[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c;
    g.stamp[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r;
  }
  if constexpr (EPI == 1) {   // bias + relu-sum + mask instead of C (md_epi_bias_relu); the mask goes through A0, 32 KiB at most
    constexpr int NPASS = (BM * BN > BM * BK * 4) ? (BM * BN) / (BM * BK * 4) : 1;
    __syncthreads();
    md_epi_bias_relu<BM, BN, WM, WN, NPASS>(acc, g, m0, n0, reinterpret_cast<uint32_t *>(A0), B0);
    return;
  }
  if constexpr (CT) {
    // C^T addressing of a swapped TT product: rows are the unit-stride axis. A lane holds four consecutive rows of ONE column, so
    // stored as they stand a wave-store is 32 columns x 32 B — pieces of lines, which drain slowly (the kernel lasted 1041 us against
    // NN's 942 with dispatches serialised). Each 32 x 32 accumulator tile is transposed through a wave-private LDS patch instead
    // ([column][row], rows padded to 36 words: 16-B aligned, conflict-light) and leaves as whole 128-B row segments: eight lanes per
    // C^T row, eight rows per store.
    __shared__ __attribute__((aligned(16))) float Tt[WM * WN][32][36];
    float(*T)[36] = Tt[wave];
    const int rl = lane >> 3, cl = (lane & 7) * 4;   // read side: column rl + 8k of the tile, rows cl .. cl + 3
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4 *>(&T[l32][8 * q + 4 * h]) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        __builtin_amdgcn_wave_barrier();   // (the LDS queue of a wave is in order: the reads below see the writes above)
        const int64_t row = m0 + wm * (WTM * 32) + i * 32 + cl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int64_t col = n0 + wn * (WTN * 32) + j * 32 + rl + 8 * k;
          *reinterpret_cast<f32x4 *>(C + row + col * g.c_ns) = *reinterpret_cast<const f32x4 *>(&T[rl + 8 * k][cl]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int64_t col = n0 + wn * (WTN * 32) + j * 32 + l32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * (WTM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (RAGGED == 0 || (row < g.M && col < g.N)) C[row * g.c_ms + col * g.c_ns] = acc[i][j][r];
      }
    }
}

// Repack of a misaligned operand (HipExec::gemm): rows x cols with unit stride along cols and ANY row stride / start -> rows padded to
// `ld`. A block copies 1024 columns of four rows: coalesced 4-B loads (the source rows start anywhere), 4-B stores.
__global__ void __launch_bounds__(256) k_repack_rows(const float *__restrict__ src, int64_t rows, int64_t cols, int64_t row_stride, float *__restrict__ dst,
                                                     int64_t ld) {
  const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x, r0 = (int64_t)blockIdx.y * 4;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int64_t r = r0 + rr;
    if (r >= rows) break;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t c = c0 + 256 * k;
      v[k] = c < cols ? src[r * row_stride + c] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t c = c0 + 256 * k;
      if (c < cols) dst[r * ld + c] = v[k];
    }
  }
}

// ---- plain tiled kernel: any dtype, any strides (f64 / ints / tiny problems) ------
template <class T>
__global__ void __launch_bounds__(256) k_gemm_generic(MdGemm g) {
  __shared__ T As[16][17];
  __shared__ T Bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t bz = blockIdx.z;
  const T *A = (const T *)g.a + bz * g.a_bs;
  const T *B = (const T *)g.b + bz * g.b_bs;
  T *C = (T *)g.c + bz * g.c_bs;
  const int64_t row = (int64_t)blockIdx.y * 16 + ty, col = (int64_t)blockIdx.x * 16 + tx;
  T acc = (T)0;
  for (int64_t k0 = 0; k0 < g.K; k0 += 16) {
    const int64_t ka = k0 + tx, kb = k0 + ty;
    As[ty][tx] = (row < g.M && ka < g.K) ? A[row * g.a_ms + ka * g.a_ks] : (T)0;
    Bs[ty][tx] = (kb < g.K && col < g.N) ? B[kb * g.b_ks + col * g.b_ns] : (T)0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if constexpr (md_is_float<T>::value) acc = fma(As[ty][k], Bs[k][tx], acc);
      else acc = BAdd::apply(acc, BMul::apply(As[ty][k], Bs[k][tx]));
    }
    __syncthreads();
  }
  if (row < g.M && col < g.N) C[row * g.c_ms + col * g.c_ns] = acc;
}

// C[b][m][n] = sum over splits (in split order: deterministic) of the partial products
__global__ void __launch_bounds__(MD_BLOCK) k_gemm_splitk_sum(const float *__restrict__ part, int splits, int64_t batch, int64_t M, int64_t N,
                                                             float *C, int64_t c_bs, int64_t c_ms, int64_t c_ns) {
  const int64_t total = batch * M * N, gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
    float acc = part[i];
    for (int s = 1; s < splits; ++s) acc += part[(int64_t)s * total + i];
    const int64_t b = i / (M * N), r = i - b * (M * N), m = r / N, n = r - m * N;
    C[b * c_bs + m * c_ms + n * c_ns] = acc;
  }
}

// 64 B of zeros in device memory for the ragged direct-to-LDS kernels (GemmArgs::zero); allocated once, never freed
static const float *md_zero_block() {
  static const float *z = [] {
    void *p = nullptr;
    if (hipMalloc(&p, 64) != hipSuccess || hipMemset(p, 0, 64) != hipSuccess) return (const float *)nullptr;
    return (const float *)p;
  }();
  return z;
}

// Three LDS buffers (DMA two k-tiles ahead, counted vmcnt at the k-tile boundary) for the 128-row direct-to-LDS tiles: when the grid
// gives a CU one block at most — the third buffer's 32 KiB cost nothing then, and such grids (2048^3, the 1024-row shards of an 8-rank
// cfg4, the all-reduce panels) are the ones whose k-tile (1.7 us) is about one DMA round trip: 2048^3 NN 132.9 -> 139.0, NT 132.7 ->
// 140.1 TFLOP/s (profiles/r3_gemm_nbuf_ab.log). Larger grids keep two buffers and two blocks per CU. option gemm_nbuf = 2 / 3 forces
// (A/B runs, exactness tests).
static bool md_gemm_nbuf3(int64_t blocks, int64_t K, int bk) {
  if (K < 2 * bk) return false;
  if (const int64_t f = md_opt(MD_OPT_GEMM_NBUF)) return f == 3;
  return blocks <= MD_NUM_CUS;
}

// Launch of a main GEMM kernel: with events attached (mdhip_event_attach_next, bench.py) the dispatch itself carries the start / stop
// timestamps — no marker packets around the kernel.
static void md_gemm_launch(void (*kernel)(GemmArgs), dim3 grid, int threads, const GemmArgs &ga) {
  hipEvent_t e0, e1;
  if (md_prof_take(&e0, &e1)) hipExtLaunchKernelGGL(kernel, grid, dim3((unsigned)threads), 0, md_stream(), e0, e1, 0, ga);
  else hipLaunchKernelGGL(kernel, grid, dim3((unsigned)threads), 0, md_stream(), ga);
}

// diagnostic only (MDHIP_GEMM_STAMP=1): one persistent stamp buffer; what the last stamped launch wrote is printed at exit
struct StampDump {
  static constexpr size_t kMax = 8192;
  unsigned long long *dev = nullptr;
  int bm = 0, bn = 0, bk = 0;
  size_t blocks = 0;
  static StampDump &get() { static StampDump d; return d; }
  unsigned long long *buffer() {
    if (!dev) (void)hipMalloc((void **)&dev, kMax * 16);
    return dev;
  }
  void note(int m, int n, int k, size_t b) { bm = m; bn = n; bk = k; blocks = b; }
  ~StampDump() {
    if (!dev || !blocks) return;
    std::vector<unsigned long long> h(blocks * 2);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return;
    std::vector<double> ghz, us;
    for (size_t b = 0; b < blocks; ++b)
      if (h[2 * b + 1]) { ghz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); us.push_back((double)h[2 * b + 1] * 0.01); }
    std::sort(ghz.begin(), ghz.end());
    std::sort(us.begin(), us.end());
    if (!ghz.empty())
      fprintf(stderr, "[mdhip] last gemm %dx%dx%d: in-kernel clock median %.3f GHz (min %.3f, max %.3f) over %zu blocks; block main loop median %.1f us (min %.1f, p10 %.1f, p90 %.1f, max %.1f)\n",
              bm, bn, bk, ghz[ghz.size() / 2], ghz.front(), ghz.back(), ghz.size(), us[us.size() / 2], us.front(), us[us.size() / 10], us[us.size() * 9 / 10], us.back());
  }
};

template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int SCHED = 0, int EPI = 0>
static int launch_cfg(GemmArgs ga, int64_t batch, bool edge) {
  ga.tiles_m = (int)((ga.M + BM - 1) / BM);
  ga.tiles_n = (int)((ga.N + BN - 1) / BN);
  ga.vec_ok = edge ? 0 : 1;  // `edge` on entry = operands not 16-B aligned
  {
    const int sh = (int)md_opt(MD_OPT_GEMM_SUPER);
    ga.super_h = (sh > 1 && ga.tiles_m >= sh && ga.tiles_n >= 8) ? sh : 0;
  }
  const bool aligned = !edge;   // (on entry `edge` only says whether the operands are 16-B aligned)
  edge = edge || (ga.M % BM) || (ga.N % BN) || (ga.K % BK);
  if constexpr (EPI != 0) {  // whole aligned tiles only; the caller falls back to the plain product otherwise
    if (edge || batch != 1) return MDHIP_EVALUE;
    dim3 grid((unsigned)(ga.tiles_m * ga.tiles_n), 1, 1);
    md_gemm_launch(k_gemm_f32_mfma<BM, BN, BK, WM, WN, A_KC, B_KC, false, false, SCHED, EPI>, grid, 64 * WM * WN, ga);
    return MD_LAUNCH_CHECK("matmul(f32 mfma, bias+relu epilogue)");
  }
  // split-K: a grid that cannot give every CU a block (skinny M or N with a long K, e.g. a
  // 64 x 4096 batch through a 4096 x 4096 layer) is widened along k; >= 512 k per split
  const int64_t tiles = (int64_t)ga.tiles_m * ga.tiles_n * batch;
  int64_t splits = 1;
  const int splitk_mode = (int)md_opt(MD_OPT_GEMM_SPLITK);
  if (splitk_mode && BM == 64 && BN == 64 && tiles < MD_NUM_CUS && ga.K >= 1024) {  // (pick_cfg sends such shapes to 64x64)
    splits = (2 * MD_NUM_CUS + tiles - 1) / tiles;
    if (splits > ga.K / 512) splits = ga.K / 512;
    if (splits > 256) splits = 256;   // (64 until round 4: a single 64 x 64 output under k = 10^6 ran on 64 blocks, 13 TFLOP/s)
  }
  ga.k_chunk = ga.K;
  ga.c_split = 0;
  void *partial = nullptr;
  GemmArgs out = ga;
  if (splits > 1) {
    ga.k_chunk = ((ga.K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (ga.K + ga.k_chunk - 1) / ga.k_chunk;
    MD_TRY(mdhip_alloc((size_t)(splits * batch * ga.M * ga.N) * sizeof(float), &partial));
    ga.C = (float *)partial;
    ga.c_bs = ga.M * ga.N; ga.c_ms = ga.N; ga.c_ns = 1;
    ga.c_split = batch * ga.M * ga.N;
    edge = edge || (ga.k_chunk % BK) || (ga.K % ga.k_chunk % BK);
  }
  dim3 grid((unsigned)(ga.tiles_m * ga.tiles_n), (unsigned)splits, (unsigned)batch);
  if constexpr (BM == 64 && BN == 64) {
    if (splits > 1) {
      if (edge) k_gemm_f32_mfma<BM, BN, BK, WM, WN, A_KC, B_KC, true, true><<<grid, 64 * WM * WN, 0, md_stream()>>>(ga);
      else k_gemm_f32_mfma<BM, BN, BK, WM, WN, A_KC, B_KC, false, true><<<grid, 64 * WM * WN, 0, md_stream()>>>(ga);
    }
  }
  const bool stamp = md_opt(MD_OPT_GEMM_STAMP) == 1;
  if (stamp && splits == 1 && !edge && (size_t)grid.x * grid.z <= StampDump::kMax) {
    ga.stamp = StampDump::get().buffer();   // persistent: no sync behind the launch, the LAST launch's stamps are printed at exit
    StampDump::get().note(BM, BN, BK, (size_t)grid.x * grid.z);
  }
  if (splits == 1) {
    bool ragged_dma = false;
    if constexpr (!A_KC && !B_KC && SCHED != 0 && BK >= 32) {
      // ragged TN: still direct to LDS when every 16-B piece lies wholly inside or outside the operands (M, N multiples of 4)
      ragged_dma = edge && aligned && md_opt(MD_OPT_GEMM_GLDS) != 0 && ga.a_ms == 1 && ga.b_ns == 1 && !ga.stamp && (ga.M % 4 == 0 || ga.pad_m) && (ga.N % 4 == 0 || ga.pad_n) && ga.a_ks > 0 && ga.b_ks > 0 &&
                   (ga.zero = md_zero_block()) != nullptr;
      if (ragged_dma) {
        // whole k-tiles on the 128-row tiles: predicated lanes + scalar-base addresses (4100 x 4096 x 4100: 104-108 -> 112-124 TFLOP/s);
        // the 256x256 tile keeps the select form — the predicate's branch around each DMA splits its one scheduling region
        // (4000^3: 133-137 against 130 with the predicate)
        // (the predicated form builds its addresses from 32-bit lane offsets: operand strides below 2^26 elements, as in the whole-tile branch)
        if (ga.K % BK == 0 && BM <= 128 && ga.a_ks < (1ll << 26) && ga.b_ks < (1ll << 26)) md_gemm_launch(k_gemm_f32_tn_glds<BM, BN, BK, WM, WN, 2>, grid, 64 * WM * WN, ga);
        else md_gemm_launch(k_gemm_f32_tn_glds<BM, BN, BK, WM, WN, 1>, grid, 64 * WM * WN, ga);
      }
    }
    if (ragged_dma) {
    } else if (edge) md_gemm_launch(k_gemm_f32_mfma<BM, BN, BK, WM, WN, A_KC, B_KC, true>, grid, 64 * WM * WN, ga);
    else {
      bool glds = false;
      if constexpr (!A_KC && !B_KC && SCHED != 0 && BK >= 32) {
        // TN with whole aligned tiles and 32-deep k-tiles: both operands go direct to LDS (option gemm_glds = 0 keeps the
        // register-staged kernel: A/B runs). Same-box A/B, profiles/r2_gemm_glds_ab.log: 256x256x32 at
        // 4096^3 140.1 -> 141.9 TFLOP/s, the 8-wave 128x128x32 at 2048^3 119.0 -> 127.8; the 16-deep 256x128 tile LOSES
        // (135.7 -> 130.8: its k-tile is too short for a one-tile-ahead DMA) and keeps its registers.
        glds = md_opt(MD_OPT_GEMM_GLDS) != 0 && ga.a_ms == 1 && ga.b_ns == 1 && !ga.stamp && ga.a_ks > 0 && ga.b_ks > 0 && ga.a_ks < (1ll << 26) && ga.b_ks < (1ll << 26);   // (32-bit lane offsets)
        if (glds) {
          if constexpr (3 * (BM + BN) * BK * 4 <= 128 * 1024) {
            if (md_gemm_nbuf3((int64_t)grid.x * grid.z, ga.K, BK)) {
              md_gemm_launch(k_gemm_f32_tn_glds<BM, BN, BK, WM, WN, 0, 3>, grid, 64 * WM * WN, ga);
              return MD_LAUNCH_CHECK("matmul(f32 mfma, direct-to-LDS TN, 3 buffers)");
            }
          }
          md_gemm_launch(k_gemm_f32_tn_glds<BM, BN, BK, WM, WN>, grid, 64 * WM * WN, ga);
        }
      }
      if (!glds) md_gemm_launch(k_gemm_f32_mfma<BM, BN, BK, WM, WN, A_KC, B_KC, false, false, SCHED>, grid, 64 * WM * WN, ga);
    }
  }
  if (splits > 1) {
    k_gemm_splitk_sum<<<md_grid_for(batch * ga.M * ga.N), MD_BLOCK, 0, md_stream()>>>((const float *)partial, (int)splits, batch, ga.M, ga.N,
                                                                                   out.C, out.c_bs, out.c_ms, out.c_ns);
    int rc = MD_LAUNCH_CHECK("matmul(f32 mfma, split-k)");
    mdhip_free(partial);
    return rc;
  }
  return MD_LAUNCH_CHECK("matmul(f32 mfma)");
}

// (128x128x32, 8-wave 256x128 and 256x256 tiles were measured and dropped: profiles/r1_gemm_tile_ab.log, r2_gemm_small_grid_ab.log)
enum { CFG_128x128x16 = 0, CFG_64x64x16, CFG_128x64x16, CFG_256x128x16, CFG_256x256x32, CFG_128x128x32, CFG_128x64x32, CFG_128x128_W8, CFG_COUNT };

// `est` (optional): the model's time for the chosen tile, in units of 512 K / 1e12 seconds (rounds x BM x BN / TFLOP/s)
static int pick_cfg(const GemmArgs &ga, int64_t batch, bool vector_staged, bool dma_ok = false, double *est = nullptr) {
  if (est) *est = 64.0 * 64.0 / 120.0;   // (the early returns: one round of the small tile, split-K or not)
  {  // experiments / exactness tests only (option gemm_cfg)
    const int64_t v = md_opt(MD_OPT_GEMM_CFG);
    if (v >= 0 && v < CFG_COUNT) return (int)v;
  }
  // Cost model over the production tiles: a CU works through ceil(tiles / CUs) tiles of BM*BN outputs at the rate measured
  // for that tile on full grids (profiles/r1_gemm_tile_ab.log, r2_gemm_small_grid_ab.log, r2_gemm_glds_ab.log) — big tiles
  // win on efficiency, small tiles on the partial last round of a grid that does not divide evenly (4097 rows: 3 rounds of
  // 256x128 against 17 of 64x64, i.e. 0.80x the time). `dma_ok`: operands aligned for the direct-to-LDS kernels, which
  // take whole tiles and 32-deep k-tiles only; their rates apply to the candidates that divide the problem.
  if (((ga.M + 63) / 64) * ((ga.N + 63) / 64) * batch < MD_NUM_CUS && ga.K >= 1024) return CFG_64x64x16;  // split-K candidates
  struct Cand { int cfg, bm, bn; double tf, tf_dma; };   // tf_dma = 0: no direct-to-LDS form of this tile
  static const Cand cands[] = {{CFG_256x256x32, 256, 256, 0.0, 149.0}, {CFG_256x128x16, 256, 128, 139.0, 0.0}, {CFG_128x128x16, 128, 128, 133.0, 143.0},
                               {CFG_128x128_W8, 128, 128, 133.0, 143.5}, {CFG_128x64x16, 128, 64, 126.0, 0.0}, {CFG_64x64x16, 64, 64, 120.0, 0.0}};
  static const Cand cands_tn[] = {{CFG_256x256x32, 256, 256, 140.0, 150.0}, {CFG_256x128x16, 256, 128, 138.0, 0.0}, {CFG_128x128x32, 128, 128, 132.0, 146.0},
                                  {CFG_128x128_W8, 128, 128, 134.0, 146.0}, {CFG_128x64x32, 128, 64, 126.7, 138.0}, {CFG_64x64x16, 64, 64, 120.0, 0.0}};
  int best = CFG_64x64x16;
  double best_t = 1e300;
  for (int ci = 0; ci < 6; ++ci) {
    const Cand &c = vector_staged ? cands_tn[ci] : cands[ci];
    const bool dma = dma_ok && c.tf_dma > 0.0;   // (ragged sizes run the same kernels with zero-filled edges: launch_mfma checks what they need)
    const double tf = dma ? c.tf_dma : c.tf;
    if (tf <= 0.0) continue;
    const int64_t tiles = ((ga.M + c.bm - 1) / c.bm) * ((ga.N + c.bn - 1) / c.bn) * batch;
    const double rounds = (double)((tiles + MD_NUM_CUS - 1) / MD_NUM_CUS);
    double t = rounds * c.bm * c.bn / tf;
    // a lone four-wave block per CU (one wave per SIMD) cannot keep the matrix pipe fed — except the 256x256 tile, whose
    // rate was measured that way (16 MFMAs per step and wave), and the eight-wave tile
    // (and the four-wave 128x128x32 direct-to-LDS tile of the TN layout: 127 against 119 TFLOP/s for the eight-wave one at 2048^3)
    if (tiles <= MD_NUM_CUS && c.cfg != CFG_128x128_W8 && c.cfg != CFG_256x256x32 && !(vector_staged && dma && c.cfg == CFG_128x128x32)) t /= 0.8;
    if (c.cfg == CFG_256x256x32 && tiles < MD_NUM_CUS) continue;   // (half-empty chip: never the best choice)
    // the register-staged kernel's guarded edge variant runs ~15 % below its whole-tile rate (4000^3: 112-115 against 133-139);
    // the direct-to-LDS kernels pay nothing for a ragged edge (zero-filled DMA lanes)
    if (!dma && ((ga.M % c.bm) || (ga.N % c.bn) || (ga.K % 16))) t /= 0.85;
    if (t < best_t * 0.999) { best_t = t; best = c.cfg; }   // ties go to the larger tile (listed first)
  }
  if (est && best_t < 1e299) *est = best_t;
  return best;
}

// NN / NT with whole aligned tiles: the direct-to-LDS kernel for k-contiguous operands (k_gemm_f32_kc_glds); -1 = not applicable
template <int BM, int BN, int WM, int WN, bool B_KC, int EPI = 0, int BK = 32>
static int launch_kc_glds(GemmArgs ga, int64_t batch, bool edge) {
  if (edge || ga.a_ks != 1 || (B_KC ? ga.b_ks != 1 : ga.b_ns != 1)) return -1;
  // (the per-lane part of a DMA address is a 32-bit byte offset of up to 15 rows / 3 k-rows: strides below 2^26 elements)
  if (ga.a_ms >= (1ll << 26) || (B_KC ? ga.b_ns : ga.b_ks) >= (1ll << 26) || ga.a_ms < 0 || ga.b_ns < 0 || ga.b_ks < 0) return -1;
  const bool ragged = (ga.M % BM) || (ga.N % BN) || (ga.K % BK);
  if (ragged) {   // every 16-B piece wholly inside or outside the operands: K (k-contiguous operands) and N (NN's B) multiples of 4
    if (EPI != 0 || (ga.K % 4) || (!B_KC && (ga.N % 4)) || (ga.zero = md_zero_block()) == nullptr) return -1;
  }
  ga.tiles_m = (int)((ga.M + BM - 1) / BM);
  ga.tiles_n = (int)((ga.N + BN - 1) / BN);
  const int sh = (int)md_opt(MD_OPT_GEMM_SUPER);
  ga.super_h = (sh > 1 && ga.tiles_m >= sh && ga.tiles_n >= 8) ? sh : 0;
  dim3 grid((unsigned)(ga.tiles_m * ga.tiles_n), 1, (unsigned)batch);
  const bool stamp = md_opt(MD_OPT_GEMM_STAMP) == 1;
  if (stamp && (size_t)grid.x * grid.z <= StampDump::kMax) {
    ga.stamp = StampDump::get().buffer();
    StampDump::get().note(BM, BN, BK, (size_t)grid.x * grid.z);
  }
  if constexpr (EPI == 0) {
    if (ragged) {
      if (ga.K % BK == 0 && BM <= 128) md_gemm_launch(k_gemm_f32_kc_glds<BM, BN, BK, WM, WN, B_KC, 0, 2>, grid, 64 * WM * WN, ga);   // (as in launch_cfg's TN branch)
      else md_gemm_launch(k_gemm_f32_kc_glds<BM, BN, BK, WM, WN, B_KC, 0, 1>, grid, 64 * WM * WN, ga);
      return MD_LAUNCH_CHECK("matmul(f32 mfma, direct-to-LDS, ragged)");
    }
  }
  if constexpr (EPI == 0 && BM <= 128) {
    if (md_gemm_nbuf3((int64_t)grid.x * grid.z, ga.K, BK)) {
      if constexpr (!B_KC) {
        if (ga.c_vec_rows) {
          md_gemm_launch(k_gemm_f32_kc_glds<BM, BN, BK, WM, WN, B_KC, 0, 0, 3, true>, grid, 64 * WM * WN, ga);
          return MD_LAUNCH_CHECK("matmul(f32 mfma, direct-to-LDS, 3 buffers, C^T stores)");
        }
      }
      md_gemm_launch(k_gemm_f32_kc_glds<BM, BN, BK, WM, WN, B_KC, 0, 0, 3>, grid, 64 * WM * WN, ga);
      return MD_LAUNCH_CHECK("matmul(f32 mfma, direct-to-LDS, 3 buffers)");
    }
  }
  if constexpr (EPI == 0 && !B_KC) {
    if (ga.c_vec_rows) {
      md_gemm_launch(k_gemm_f32_kc_glds<BM, BN, BK, WM, WN, B_KC, 0, 0, 2, true>, grid, 64 * WM * WN, ga);
      return MD_LAUNCH_CHECK("matmul(f32 mfma, direct-to-LDS, C^T stores)");
    }
  }
  md_gemm_launch(k_gemm_f32_kc_glds<BM, BN, BK, WM, WN, B_KC, EPI>, grid, 64 * WM * WN, ga);
  return MD_LAUNCH_CHECK("matmul(f32 mfma, direct-to-LDS)");
}

template <bool A_KC, bool B_KC>
static bool md_gemm_dma_ok(const GemmArgs &ga, bool edge) {
  // direct-to-LDS kernels: aligned operands whose vector axis has stride 1 (option gemm_glds = 0 keeps
  // the register-staged kernels: A/B runs and tests)
  return md_opt(MD_OPT_GEMM_GLDS) != 0 && !edge && !ga.stamp && (A_KC ? ga.a_ks == 1 : ga.a_ms == 1) && (B_KC ? ga.b_ks == 1 : ga.b_ns == 1) &&
         (A_KC || !B_KC) &&   // (A row-contiguous with B k-contiguous — "TT" — has no such kernel: HipExec::gemm swaps it into NN)
         // sizes the tiles do not divide: every 16-B piece must lie wholly inside or outside its operand
         ((A_KC || B_KC) ? ga.K % 4 == 0 : true) && (A_KC || ga.M % 4 == 0 || ga.pad_m) && (B_KC || ga.N % 4 == 0 || ga.pad_n);
}

template <bool A_KC, bool B_KC>
static int launch_mfma(const GemmArgs &ga, int64_t batch, bool edge) {
  const bool dma_ok = md_gemm_dma_ok<A_KC, B_KC>(ga, edge);
  const int cfg = pick_cfg(ga, batch, !A_KC && !B_KC, dma_ok);
  if constexpr (A_KC) {
    if (dma_ok) {
      int rc = -1;
      if (cfg == CFG_256x256x32) rc = launch_kc_glds<256, 256, 2, 2, B_KC>(ga, batch, edge);
      else if (cfg == CFG_128x128_W8) rc = launch_kc_glds<128, 128, 2, 4, B_KC>(ga, batch, edge);   // (64-deep k-tiles: no gain, before and after the addressing change)
      else if (cfg == CFG_128x128x32 || cfg == CFG_128x128x16) rc = launch_kc_glds<128, 128, 2, 2, B_KC>(ga, batch, edge);
      if (rc >= 0) return rc;
    }
  }
  switch (cfg) {
    case CFG_64x64x16: return launch_cfg<64, 64, 16, 2, 2, A_KC, B_KC, 0>(ga, batch, edge);   // (2 MFMAs per step: nothing to interleave with; measured 6 % slower)
    case CFG_128x64x16: return launch_cfg<128, 64, 16, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
    case CFG_256x128x16: return launch_cfg<256, 128, 16, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
    // deeper k-tiles for the small tiles of the TN layout: a k-tile of a small tile is too short to cover the
    // global-load latency of the tile staged two ahead (2048^3 TN: 106 -> 117 TFLOP/s); the k-contiguous
    // layouts are held back by their transposing LDS stores instead and gain nothing from it
    case CFG_128x128x32:
      if constexpr (!A_KC && !B_KC) return launch_cfg<128, 128, 32, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
      else return launch_cfg<128, 128, 16, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
    case CFG_128x64x32:
      if constexpr (!A_KC && !B_KC) return launch_cfg<128, 64, 32, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
      else return launch_cfg<128, 64, 16, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
    case CFG_256x256x32:  // both operands staged with vector LDS stores (TN): one block per CU, half the barriers
      if constexpr (!A_KC && !B_KC) return launch_cfg<256, 256, 32, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
      else return launch_cfg<256, 128, 16, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
    // eight waves on ONE 128x128 tile: two waves per SIMD from a single block per CU — for grids of about one tile per CU
    // (2048^3, the 1024-row shards of cfg4 at 8 ranks), where the four-wave tiles either leave every SIMD one wave
    // (128x128) or pay for twice the operand traffic (two 128x64 blocks): 2048^3 109-114 -> 118-123 TFLOP/s,
    // 1024x4096x4096 114-118 -> 122-126 (profiles/r2_gemm_small_grid_ab.log)
    case CFG_128x128_W8:
      if constexpr (!A_KC && !B_KC) return launch_cfg<128, 128, 32, 2, 4, A_KC, B_KC, 1>(ga, batch, edge);
      else return launch_cfg<128, 128, 16, 2, 4, A_KC, B_KC, 1>(ga, batch, edge);
    default: return launch_cfg<128, 128, 16, 2, 2, A_KC, B_KC, 1>(ga, batch, edge);
  }
}


// A product whose M and / or N are a few rows past a multiple of the big tile (x.T of a 4097-column matrix: 4097 x 4096 x 4100)
// pays for its ragged edge with a whole extra ROUND of tiles in a single launch (17 x 17 tiles of 256^2 for 256.3 tiles of work), or
// runs everything on small tiles. Peeled instead: the aligned main block [0, Mm) x [0, Nm) on the whole-tile kernels, the bottom strip
// [Mm, M) x [0, N) and the right strip [0, Mm) x [Nm, N) as two small products (thin strips take the split-K route) — three launches
// on one stream writing disjoint parts of C. Chosen by the same cost model as the tiles, when it beats the single launch by > 3 %.
template <bool A_KC, bool B_KC>
static int launch_mfma_peeled(const GemmArgs &ga, int64_t batch, bool edge) {
  const int mode = (int)md_opt(MD_OPT_GEMM_PEEL);   // 0 never, 1 by the model, 2 whenever there is an edge to peel
  const int64_t G = 256;
  const int64_t Mm = ga.M / G * G, Nm = ga.N / G * G, Mr = ga.M - Mm, Nr = ga.N - Nm;
  if (mode == 0 || edge || ga.stamp || (Mr == 0 && Nr == 0) || Mm == 0 || Nm == 0 || ga.K % 32 || ga.bias) return launch_mfma<A_KC, B_KC>(ga, batch, edge);
  auto sub = [&](int64_t m0, int64_t m1, int64_t n0, int64_t n1) {
    GemmArgs x = ga;
    x.A = ga.A + m0 * ga.a_ms; x.B = ga.B + n0 * ga.b_ns; x.C = ga.C + m0 * ga.c_ms + n0 * ga.c_ns;
    x.M = m1 - m0; x.N = n1 - n0;
    if (x.c_vec_rows && (x.M % 4 || ((uintptr_t)x.C & 15))) x.c_vec_rows = 0;
    return x;
  };
  const GemmArgs main_blk = sub(0, Mm, 0, Nm), bottom = sub(Mm, ga.M, 0, ga.N), right = sub(0, Mm, Nm, ga.N);
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  if (!al16(bottom.A) || !al16(right.B)) return launch_mfma<A_KC, B_KC>(ga, batch, edge);   // (cannot happen for aligned operands: Mm, Nm are multiples of 256)
  if (mode == 1) {
    double t_one = 0, t_main = 0, t_b = 0, t_r = 0;
    pick_cfg(ga, batch, !A_KC && !B_KC, md_gemm_dma_ok<A_KC, B_KC>(ga, edge), &t_one);
    pick_cfg(main_blk, batch, !A_KC && !B_KC, md_gemm_dma_ok<A_KC, B_KC>(main_blk, edge), &t_main);
    if (Mr) pick_cfg(bottom, batch, !A_KC && !B_KC, md_gemm_dma_ok<A_KC, B_KC>(bottom, edge), &t_b);
    if (Nr) pick_cfg(right, batch, !A_KC && !B_KC, md_gemm_dma_ok<A_KC, B_KC>(right, edge), &t_r);
    const double launch = 6e-6 * 1e12 / (512.0 * (double)ga.K);   // ~6 us per extra launch (prologue / epilogue of a lone small kernel), in model units
    // the ragged single launch runs ~8 % under the model (4097 x 4096 x 4100: 112 TFLOP/s measured against 123 modelled)
    if (t_main + t_b + t_r + launch * ((Mr != 0) + (Nr != 0)) > 0.97 * (t_one / 0.92)) return launch_mfma<A_KC, B_KC>(ga, batch, edge);
  }
  auto strip = [&](const GemmArgs &x) {   // a strip of up to eight rows / columns is a skinny product (skinny.hip: one read of the big operand)
    MdGemm sg;
    sg.batch = batch; sg.M = x.M; sg.N = x.N; sg.K = x.K;
    sg.a = x.A; sg.b = x.B; sg.c = x.C;
    sg.a_bs = x.a_bs; sg.a_ms = x.a_ms; sg.a_ks = x.a_ks;
    sg.b_bs = x.b_bs; sg.b_ks = x.b_ks; sg.b_ns = x.b_ns;
    sg.c_bs = x.c_bs; sg.c_ms = x.c_ms; sg.c_ns = x.c_ns;
    const int rc = md_gemm_skinny(sg, MDHIP_F32);
    return rc >= 0 ? rc : launch_mfma<A_KC, B_KC>(x, batch, edge);
  };
  int rc = launch_mfma<A_KC, B_KC>(main_blk, batch, edge);
  if (rc == MDHIP_OK && Mr) rc = strip(bottom);
  if (rc == MDHIP_OK && Nr) rc = strip(right);
  return rc;
}


// ======================= f64: v_mfma_f64_16x16x4_f64 ===========================================
// Same staging scheme as the f32 kernel (k-major LDS double buffer, registers hold the tile in
// flight, fragment reads one step ahead), 16x16x4 MFMA tiles: lane l feeds A[l&15][k=l>>4] /
// B[k=l>>4][l&15]; the accumulator holds C[(l>>4)+4r][l&15], r=0..3 (the f64 map differs from
// every other MFMA shape, cdna_hip_programming.md "Fragment layout"). Bound: 78.6 TFLOP/s.
// Serves the reference's own tests, which run in float64 (tests/test_ops.py:311-323, (10,30)@(30,20)
// and up) and any f64 user; the fp32 headline never comes here.
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int LDP64 = 18;  // LDS row pad (doubles): transposing b64 stores of a half-wave hit 64 distinct banks; fragment reads overlap by 2 lanes

struct GemmArgs64 {
  const double *A, *B;
  double *C;
  int64_t M, N, K;
  int64_t a_bs, a_ms, a_ks, b_bs, b_ks, b_ns, c_bs, c_ms, c_ns;
  int tiles_m, tiles_n;
  int vec_ok;
};

template <int ROWS, int BK, int NT, bool KC, bool EDGE>
__device__ __forceinline__ void load_tile64(const double *__restrict__ P, int64_t rs, int64_t ks, int64_t row0, int64_t k0,
                                            int64_t rows, int64_t K, f64x2 (&r)[ROWS * BK / (2 * NT)], bool vec_ok) {
  constexpr int PASSES = ROWS * BK / (2 * NT);
  constexpr int TPR = BK / 2, RPP = NT / TPR;   // KC: threads per row / rows per pass
  constexpr int TPK = ROWS / 2, KPP = NT / TPK; // !KC: threads per k-row / k-rows per pass
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < PASSES; ++i) {
    int row, k;
    if constexpr (KC) { row = t / TPR + RPP * i; k = (t % TPR) * 2; }
    else { k = t / TPK + KPP * i; row = (t % TPK) * 2; }
    if constexpr (!EDGE) {
      r[i] = *reinterpret_cast<const f64x2 *>(P + (row0 + row) * rs + (k0 + k) * ks);
    } else {
      const int64_t rr0 = row0 + row, kk0 = k0 + k;
      const bool inside = KC ? (rr0 < rows && kk0 + 1 < K) : (kk0 < K && rr0 + 1 < rows);
      if (vec_ok && inside) {
        r[i] = *reinterpret_cast<const f64x2 *>(P + rr0 * rs + kk0 * ks);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int64_t rr = rr0 + (KC ? 0 : j), kk = kk0 + (KC ? j : 0);
          r[i][j] = (rr < rows && kk < K) ? P[rr * rs + kk * ks] : 0.0;
        }
      }
    }
  }
}
template <int ROWS, int BK, int NT, bool KC>
__device__ __forceinline__ void store_tile64(double (*S)[ROWS + LDP64], const f64x2 (&r)[ROWS * BK / (2 * NT)]) {
  constexpr int PASSES = ROWS * BK / (2 * NT);
  constexpr int TPR = BK / 2, RPP = NT / TPR, TPK = ROWS / 2, KPP = NT / TPK;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < PASSES; ++i) {
    if constexpr (KC) {
      const int row = t / TPR + RPP * i, k = (t % TPR) * 2;
      S[k][row] = r[i][0];
      S[k + 1][row] = r[i][1];
    } else {
      const int k = t / TPK + KPP * i, row = (t % TPK) * 2;
      *reinterpret_cast<f64x2 *>(&S[k][row]) = r[i];
    }
  }
}

template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, bool EDGE>
__global__ void __launch_bounds__(64 * WM * WN) k_gemm_f64_mfma(GemmArgs64 g) {
  constexpr int NT = 64 * WM * WN;
  constexpr int WTM = BM / (16 * WM), WTN = BN / (16 * WN);
  __shared__ double As[2][BK][BM + LDP64];
  __shared__ double Bs[2][BK][BN + LDP64];
  const int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t bz = blockIdx.z;
  const double *A = g.A + bz * g.a_bs;
  const double *B = g.B + bz * g.b_bs;
  double *C = g.C + bz * g.c_bs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l16 = lane & 15, h = lane >> 4;
  const bool vec_ok = g.vec_ok != 0;

  f64x4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  f64x2 ra[BM * BK / (2 * NT)], rb[BN * BK / (2 * NT)];
  const int64_t nk = (g.K + BK - 1) / BK;
  load_tile64<BM, BK, NT, A_KC, EDGE>(A, g.a_ms, g.a_ks, m0, 0, g.M, g.K, ra, vec_ok);
  load_tile64<BN, BK, NT, B_KC, EDGE>(B, g.b_ns, g.b_ks, n0, 0, g.N, g.K, rb, vec_ok);
  store_tile64<BM, BK, NT, A_KC>(As[0], ra);
  store_tile64<BN, BK, NT, B_KC>(Bs[0], rb);
  if (nk > 1) {
    load_tile64<BM, BK, NT, A_KC, EDGE>(A, g.a_ms, g.a_ks, m0, BK, g.M, g.K, ra, vec_ok);
    load_tile64<BN, BK, NT, B_KC, EDGE>(B, g.b_ns, g.b_ks, n0, BK, g.N, g.K, rb, vec_ok);
  }
  __syncthreads();

  // same schedule as the f32 kernel: branch-free k-tile body, a slice of the step's memory work behind every
  // MFMA, a tile's first fragments read right behind the barrier that publishes it
  int cur = 0;
  double fa[2][WTM], fb[2][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i) fa[0][i] = As[0][h][wm * (WTM * 16) + i * 16 + l16];
#pragma unroll
  for (int j = 0; j < WTN; ++j) fb[0][j] = Bs[0][h][wn * (WTN * 16) + j * 16 + l16];
  for (int64_t kt = 0; kt < nk; ++kt) {
    const bool more2 = kt + 2 < nk;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      const int c = (kk >> 2) & 1;
      if (kk + 4 < BK) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[c ^ 1][i] = As[cur][kk + 4 + h][wm * (WTM * 16) + i * 16 + l16];
#pragma unroll
        for (int j = 0; j < WTN; ++j) fb[c ^ 1][j] = Bs[cur][kk + 4 + h][wn * (WTN * 16) + j * 16 + l16];
      }
      if (kk == 0) store_tile64<BM, BK, NT, A_KC>(As[cur ^ 1], ra);
      if (kk == 4) store_tile64<BN, BK, NT, B_KC>(Bs[cur ^ 1], rb);
      if (kk == 8) {
        const int64_t ktl = more2 ? kt + 2 : nk - 1;
        load_tile64<BM, BK, NT, A_KC, EDGE>(A, g.a_ms, g.a_ks, m0, ktl * BK, g.M, g.K, ra, vec_ok);
        load_tile64<BN, BK, NT, B_KC, EDGE>(B, g.b_ns, g.b_ks, n0, ktl * BK, g.N, g.K, rb, vec_ok);
      }
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[c][i], fb[c][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < WTM * WTN; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (kk == 0 || kk == 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (kk == 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    __syncthreads();
    cur ^= 1;
#pragma unroll
    for (int i = 0; i < WTM; ++i) fa[0][i] = As[cur][h][wm * (WTM * 16) + i * 16 + l16];
#pragma unroll
    for (int j = 0; j < WTN; ++j) fb[0][j] = Bs[cur][h][wn * (WTN * 16) + j * 16 + l16];
  }
  // C/D of the f64 16x16 tile: col = lane&15, row = (lane>>4) + 4*r
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int64_t col = n0 + wn * (WTN * 16) + j * 16 + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * (WTM * 16) + i * 16 + h + 4 * r;
        if (!EDGE || (row < g.M && col < g.N)) C[row * g.c_ms + col * g.c_ns] = acc[i][j][r];
      }
    }
}

// ---- float64 TN on the direct-to-LDS scheme (round 3) -------------------------------------------------------------------
// Same idea as k_gemm_f32_tn_glds: `global_load_lds_dwordx4` moves 64 lanes x 16 B = 1 KiB = 128 doubles — ONE k-row of a
// 128-row tile — straight into LDS (no staging registers, no ds_write), one k-tile ahead, two separate buffers per operand with
// the buffer fixed at compile time (loop unrolled by two). 128 x 128 x 16 tiles, four waves (wave tile 64 x 64 = 4 x 4 MFMAs
// of v_mfma_f64_16x16x4_f64 per k-step, 128 accumulator registers), two blocks per CU (72 KiB of LDS each). k-rows sit
// 1152 B apart (pad 128 B: the four k of a fragment land on different banks) -> a fragment is one ds_read_b64 of S[k][row].
// Addresses: wave-uniform base (opaque to the optimiser: md_opaque_uniform) + loop-invariant 32-bit lane offset -> the DMA's
// `vN, s[base]` form (checked with scripts/isa_check.py: 16 / 16 per two k-tiles, no ds_write, no vmcnt(0) drain).
// Only the TN layout (both operands row-contiguous: the weight gradient A^T G): 4096^3 72.5-72.9 -> 73.4-75.5 TFLOP/s of 78.6.
// The k-contiguous image (16 rows x 8 k per piece, as in k_gemm_f32_kc_glds) was built and measured too: NN 67-70, NT 62-64
// against 73-74 for the register-staged kernel — a piece brings 64 B per row, half a cache line per DMA lane group — so NN / NT
// (and every ragged or unaligned shape) stay on k_gemm_f64_mfma (profiles/r3_gemm_f64.log).
constexpr int F64_TILE_LD = 128 + 16;   // doubles between the k-rows of an operand image
// k-row p (0..15) of a 128-row x 16-k operand tile whose first k is k0; `base` = operand + first row of the tile (+ batch)
__device__ __forceinline__ void glds64_krow(const double *base, int64_t ks, int64_t k0, double *S, int p, uint32_t lane_off) {
  const char *ub = md_opaque_uniform(reinterpret_cast<const char *>(base + (k0 + p) * ks));
  asm("" : "+v"(lane_off) : "s"((int)k0));   // (keeps the offset's zero-extension next to the DMA: see glds_tile_pass_u)
  __builtin_amdgcn_global_load_lds((md_gbl_void *)(ub + lane_off), (md_lds_void *)(S + p * F64_TILE_LD), 16, 0, 0);
}

__global__ void __launch_bounds__(256, 2) k_gemm_f64_tn_glds(GemmArgs64 g) {
  constexpr int BM = 128, BN = 128, BK = 16, WTM = 4, WTN = 4;
  __shared__ __attribute__((aligned(16))) double A0[BK * F64_TILE_LD];
  __shared__ __attribute__((aligned(16))) double A1[BK * F64_TILE_LD];
  __shared__ __attribute__((aligned(16))) double B0[BK * F64_TILE_LD];
  __shared__ __attribute__((aligned(16))) double B1[BK * F64_TILE_LD];
  const int nblk = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);   // XCD-aware tile order
  const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t bz = blockIdx.z;
  const double *A = g.A + bz * g.a_bs + m0;
  const double *B = g.B + bz * g.b_bs + n0;
  double *C = g.C + bz * g.c_bs;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l16 = lane & 15, h = lane >> 4;
  const uint32_t lo = (uint32_t)(lane * 16);

  f64x4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0;

  const int64_t nk = g.K / BK;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    glds64_krow(A, g.a_ks, 0, A0, i * 4 + wave, lo);
    glds64_krow(B, g.b_ks, 0, B0, i * 4 + wave, lo);
  }
  __syncthreads();

  double fa[2][WTM], fb[2][WTN];
  // fragment of k-step s: lane (l16, h) supplies k = 4 s + h
#define MD_F64_READ(BUF, s, c)                                                                                                     \
  {                                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i) fa[c][i] = ((BUF) ? A1 : A0)[(4 * (s) + h) * F64_TILE_LD + wm * 64 + i * 16 + l16]; \
    _Pragma("unroll") for (int j = 0; j < WTN; ++j) fb[c][j] = ((BUF) ? B1 : B0)[(4 * (s) + h) * F64_TILE_LD + wn * 64 + j * 16 + l16]; \
  }
  MD_F64_READ(0, 0, 0)
  auto ktile = [&](auto curc, int64_t kn) {
    constexpr int CUR = decltype(curc)::value;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = s & 1;
      if (s + 1 < 4) MD_F64_READ(CUR, s + 1, c ^ 1)
      if (s < 2) {   // the next tile's eight k-rows of this wave go out behind the first two k-steps
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          glds64_krow(A, g.a_ks, kn * BK, CUR ? A0 : A1, (2 * s + q) * 4 + wave, lo);
          glds64_krow(B, g.b_ks, kn * BK, CUR ? B0 : B1, (2 * s + q) * 4 + wave, lo);
        }
      }
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[c][i], fb[c][j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int m = 0; m < WTM * WTN; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m < 8 && s + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (s < 2 && m < 4) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
    __syncthreads();
    MD_F64_READ(CUR ^ 1, 0, 0)
  };
  int64_t kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    ktile(MdInt<0>{}, kt + 1);
    ktile(MdInt<1>{}, kt + 2 < nk ? kt + 2 : nk - 1);
  }
  if (kt < nk) ktile(MdInt<0>{}, nk - 1);
#undef MD_F64_READ
  // C/D of the f64 16x16 tile: col = lane & 15, row = (lane >> 4) + 4 r
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int64_t col = n0 + wn * 64 + j * 16 + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 64 + i * 16 + h + 4 * r;
        C[row * g.c_ms + col * g.c_ns] = acc[i][j][r];
      }
    }
}

template <int BM, int BN, bool A_KC, bool B_KC>
static int launch_f64(GemmArgs64 ga, int64_t batch, bool edge) {
  constexpr int BK = 16;
  ga.tiles_m = (int)((ga.M + BM - 1) / BM);
  ga.tiles_n = (int)((ga.N + BN - 1) / BN);
  ga.vec_ok = edge ? 0 : 1;
  edge = edge || (ga.M % BM) || (ga.N % BN) || (ga.K % BK);
  dim3 grid((unsigned)(ga.tiles_m * ga.tiles_n), 1, (unsigned)batch);
  if (edge) k_gemm_f64_mfma<BM, BN, BK, 2, 2, A_KC, B_KC, true><<<grid, 256, 0, md_stream()>>>(ga);
  else k_gemm_f64_mfma<BM, BN, BK, 2, 2, A_KC, B_KC, false><<<grid, 256, 0, md_stream()>>>(ga);
  return MD_LAUNCH_CHECK("matmul(f64 mfma)");
}
template <bool A_KC, bool B_KC>
static int launch_f64_pick(const GemmArgs64 &ga, int64_t batch, bool edge) {
  const int64_t t128 = ((ga.M + 127) / 128) * ((ga.N + 127) / 128) * batch;
  if constexpr (!A_KC && !B_KC) {   // TN: both operands row-contiguous -> direct to LDS (k_gemm_f64_tn_glds)
    const int64_t lim = 1ll << 25;
    if (md_opt(MD_OPT_GEMM_GLDS) != 0 && !edge && ga.a_ms == 1 && ga.b_ns == 1 && ga.M % 128 == 0 && ga.N % 128 == 0 && ga.K % 16 == 0 && ga.K >= 32 &&
        t128 >= MD_NUM_CUS && ga.a_ks > 0 && ga.b_ks > 0 && ga.a_ks < lim && ga.b_ks < lim) {
      GemmArgs64 g2 = ga;
      g2.tiles_m = (int)(ga.M / 128);
      g2.tiles_n = (int)(ga.N / 128);
      dim3 grid((unsigned)(g2.tiles_m * g2.tiles_n), 1, (unsigned)batch);
      k_gemm_f64_tn_glds<<<grid, 256, 0, md_stream()>>>(g2);
      return MD_LAUNCH_CHECK("matmul(f64 mfma, direct-to-LDS)");
    }
  }
  if (t128 >= 2 * MD_NUM_CUS) return launch_f64<128, 128, A_KC, B_KC>(ga, batch, edge);
  return launch_f64<64, 64, A_KC, B_KC>(ga, batch, edge);
}

struct HipExec {
  template <class T> static int gemm(const MdGemm &g_in) {
    if (g_in.batch > 65535) return md_fail(MDHIP_EVALUE, "matmul: batch extent %lld exceeds 65535", (long long)g_in.batch);
    if constexpr (md_same<T, float>::value || md_same<T, double>::value || md_same<T, int32_t>::value || md_same<T, int64_t>::value) {
      // both sides thin and k long (np.dot of two vectors): k cut over the chip, fixed-order sum of the block partials (skinny.hip)
      const int rc = md_gemm_longk(g_in, md_dtype_of<T>::value);
      if (rc >= 0) return rc;
    }
    if constexpr (md_same<T, float>::value || md_same<T, double>::value) {
      // a thin side (matrix x vector, a few columns / rows): HBM-bound streaming kernels (skinny.hip)
      const int rc = md_gemm_skinny(g_in, md_same<T, float>::value ? MDHIP_F32 : MDHIP_F64);
      if (rc >= 0) return rc;
    }
    if constexpr (md_same<T, float>::value) {
      // A product whose output is a few dozen 128 x 128 tiles under a LONG k (the weight gradient of a 1000-wide layer over a
      // 300,000-row batch: 64 tiles, a quarter of the chip, 32-52 TFLOP/s): k is cut into 2-4 ranges that run as the BATCH of one
      // launch of the ordinary kernels (a range is a batch entry: a_bs / b_bs step along k, c_bs from partial to partial), a short
      // second launch takes what the equal ranges leave over, and the partials are added in range order (k_gemm_splitk_sum).
      const int64_t t128 = ((g_in.M + 127) / 128) * ((g_in.N + 127) / 128), t64 = ((g_in.M + 63) / 64) * ((g_in.N + 63) / 64);
      // (both sides at most 2048: the row panels dp.GradSync cuts a wide weight gradient into — 512 x 4096 — keep the plain product's
      // summation order, so panelled and un-panelled gradients stay bit-identical)
      if (md_opt(MD_OPT_GEMM_SPLITK) && g_in.batch == 1 && t64 >= MD_NUM_CUS && t128 <= MD_NUM_CUS / 2 && g_in.M <= 2048 && g_in.N <= 2048 && g_in.K >= 4096 &&
          g_in.K >= 2 * (g_in.M > g_in.N ? g_in.M : g_in.N)) {
        int64_t splits = MD_NUM_CUS / t128;
        if (splits > 4) splits = 4;
        const int64_t k_chunk = g_in.K / splits / 32 * 32, k_rem = g_in.K - splits * k_chunk;
        if (splits >= 2 && k_chunk >= 1024) {
          const int64_t parts = splits + (k_rem > 0 ? 1 : 0);
          void *partial = nullptr;
          MD_TRY(mdhip_alloc((size_t)(parts * g_in.M * g_in.N) * sizeof(float), &partial));
          MdGemm g2 = g_in;
          g2.batch = splits;
          g2.K = k_chunk;
          g2.a_bs = k_chunk * g_in.a_ks;
          g2.b_bs = k_chunk * g_in.b_ks;
          g2.c = partial;
          g2.c_bs = g_in.M * g_in.N; g2.c_ms = g_in.N; g2.c_ns = 1;
          int rc = gemm<float>(g2);
          if (rc == MDHIP_OK && k_rem > 0) {
            MdGemm g3 = g2;
            g3.batch = 1;
            g3.K = k_rem;
            g3.a = (const float *)g_in.a + splits * k_chunk * g_in.a_ks;
            g3.b = (const float *)g_in.b + splits * k_chunk * g_in.b_ks;
            g3.c = (float *)partial + splits * g_in.M * g_in.N;
            rc = gemm<float>(g3);
          }
          if (rc == MDHIP_OK) {
            k_gemm_splitk_sum<<<md_grid_for(g_in.M * g_in.N), MD_BLOCK, 0, md_stream()>>>((const float *)partial, (int)parts, 1, g_in.M, g_in.N, (float *)g_in.c,
                                                                                     g_in.c_bs, g_in.c_ms, g_in.c_ns);
            rc = MD_LAUNCH_CHECK("matmul(f32, k ranges as a batch)");
          }
          mdhip_free(partial);
          return rc;
        }
      }
    }
    MdGemm g = g_in;
    bool c_rows_unit = false;
    if constexpr (md_same<T, float>::value) {
      // "TT" (A^T B^T of two row-major arrays: A unit-stride along m, B along k) has no kernel of its own: C^T = B'A' is the NN
      // product of the two STORAGES (B' = N x K k-contiguous, A' = K x M row-contiguous), so it runs on the NN direct-to-LDS
      // kernels with the operands swapped and C addressed through swapped strides; the epilogue then holds four consecutive
      // elements of a C row per lane and stores them as one 16-B vector (c_vec_rows)
      if (md_opt(MD_OPT_GEMM_TT_SWAP) != 0 && g.a_ms == 1 && g.a_ks != 1 && g.b_ks == 1 && g.b_ns != 1 && g.M > 1 && g.N > 1 && g.K > 1) {
        g.a = g_in.b; g.b = g_in.a;
        g.M = g_in.N; g.N = g_in.M;
        g.a_bs = g_in.b_bs; g.a_ms = g_in.b_ns; g.a_ks = g_in.b_ks;
        g.b_bs = g_in.a_bs; g.b_ks = g_in.a_ks; g.b_ns = g_in.a_ms;
        g.c_ms = g_in.c_ns; g.c_ns = g_in.c_ms;
        c_rows_unit = g.c_ms == 1 && (g.c_ns & 3) == 0 && (g.c_bs & 3) == 0 && ((uintptr_t)g.c & 15) == 0 && g.M % 4 == 0;
      }
      // each operand must have a unit stride along k or along its other axis
      const bool a_kc = g.a_ks == 1 || g.K == 1, a_mc = g.a_ms == 1 || g.M == 1;
      const bool b_kc = g.b_ks == 1 || g.K == 1, b_nc = g.b_ns == 1 || g.N == 1;
      const bool big = g.M * g.N >= 64 * 64 && g.K >= 8;
      if (big && (a_kc || a_mc) && (b_kc || b_nc) && g.M * g.N < (1ll << 40)) {
        GemmArgs ga{};  // (zero: no epilogue outputs, no diagnostic stamps)
        ga.A = (const float *)g.a; ga.B = (const float *)g.b; ga.C = (float *)g.c;
        ga.M = g.M; ga.N = g.N; ga.K = g.K;
        ga.a_bs = g.a_bs; ga.a_ms = g.a_ms; ga.a_ks = g.a_ks;
        ga.b_bs = g.b_bs; ga.b_ks = g.b_ks; ga.b_ns = g.b_ns;
        ga.c_bs = g.c_bs; ga.c_ms = g.c_ms; ga.c_ns = g.c_ns;
        ga.tiles_m = ga.tiles_n = 0;  // set per tile config
        ga.c_vec_rows = c_rows_unit ? 1 : 0;
        // prefer the layout that allows 16-B loads; A_KC means "vectorise A along k"
        const bool A_KC = a_kc && !(a_mc && g.a_ks != 1), B_KC = b_kc && !(b_nc && g.b_ks != 1);
        auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
        bool edge = !al16(g.a) || !al16(g.b);
        // the vector axis' partner stride must keep rows 16-B aligned
        edge = edge || ((A_KC ? g.a_ms : g.a_ks) & 3) || ((B_KC ? g.b_ns : g.b_ks) & 3) || (g.a_bs & 3) || (g.b_bs & 3);
        // Misaligned operands of a LARGE product (an odd leading dimension: x.T of a 4097-column matrix; a view that starts
        // off a 16-B boundary): one strided copy into an aligned buffer whose rows are padded to a multiple of four elements,
        // then the direct-to-LDS kernels — 4097 x 4096 x 4100 TN ran at 71 TFLOP/s on the register-staged edge kernel
        // (DESIGN §9.1). Worth it when the product is >= 50x the copy (2 M N K flop against 8 bytes per copied element).
        void *tmp_a = nullptr, *tmp_b = nullptr;
        const bool repack_on = md_opt(MD_OPT_GEMM_REPACK) != 0;
        if (edge && repack_on && g.batch == 1 && (A_KC || B_KC ? g.K % 4 == 0 : true) && g.M >= 256 && g.N >= 256 && g.K >= 256 && !ga.stamp) {
          const bool a_bad = !al16(g.a) || ((A_KC ? g.a_ms : g.a_ks) & 3), b_bad = !al16(g.b) || ((B_KC ? g.b_ns : g.b_ks) & 3);
          auto repack = [](const float *src, int64_t rows, int64_t cols, int64_t row_stride, void **tmp, int64_t *ld) -> int {
            // rows x cols, unit stride along cols -> rows x ld (ld = cols rounded up to 4; the padding stays unwritten)
            *ld = (cols + 3) & ~(int64_t)3;
            MD_TRY(mdhip_alloc((size_t)(rows * *ld) * sizeof(float), tmp));
            if (rows <= 65535 * 4) {   // (a row per blockIdx.y group: no div / mod per element — the generic strided copy took 89 us for 4096 x 4097)
              const dim3 grid((unsigned)((cols + 1023) / 1024), (unsigned)((rows + 3) / 4));
              k_repack_rows<<<grid, 256, 0, md_stream()>>>(src, rows, cols, row_stride, (float *)*tmp, *ld);
              return MD_LAUNCH_CHECK("matmul(repack)");
            }
            mdhip_array sd{}, dd{};
            sd.data = const_cast<float *>(src); sd.dtype = MDHIP_F32; sd.ndim = 2; sd.shape[0] = rows; sd.shape[1] = cols; sd.strides[0] = row_stride; sd.strides[1] = 1;
            dd = sd; dd.data = *tmp; dd.strides[0] = *ld;
            return mdhip_unary(MDHIP_U_COPY, &sd, &dd);
          };
          int rc = MDHIP_OK;
          if (a_bad) {
            int64_t ld;
            if (A_KC) { rc = repack(ga.A, g.M, g.K, g.a_ms, &tmp_a, &ld); ga.a_ms = ld; }
            else { rc = repack(ga.A, g.K, g.M, g.a_ks, &tmp_a, &ld); ga.a_ks = ld; ga.pad_m = 1; }
            if (rc == MDHIP_OK) ga.A = (const float *)tmp_a;
          }
          if (rc == MDHIP_OK && b_bad) {
            int64_t ld;
            if (B_KC) { rc = repack(ga.B, g.N, g.K, g.b_ns, &tmp_b, &ld); ga.b_ns = ld; }
            else { rc = repack(ga.B, g.K, g.N, g.b_ks, &tmp_b, &ld); ga.b_ks = ld; ga.pad_n = 1; }
            if (rc == MDHIP_OK) ga.B = (const float *)tmp_b;
          }
          if (rc != MDHIP_OK) {
            if (tmp_a) mdhip_free(tmp_a);
            if (tmp_b) mdhip_free(tmp_b);
            return rc;
          }
          edge = false;
        }
        int rc;
        if (A_KC && B_KC) rc = launch_mfma_peeled<true, true>(ga, g.batch, edge);
        else if (A_KC && !B_KC) rc = launch_mfma_peeled<true, false>(ga, g.batch, edge);
        else if (!A_KC && B_KC) rc = launch_mfma_peeled<false, true>(ga, g.batch, edge);
        else rc = launch_mfma_peeled<false, false>(ga, g.batch, edge);
        if (tmp_a) mdhip_free(tmp_a);   // stream-ordered: the next user of the block runs after the product
        if (tmp_b) mdhip_free(tmp_b);
        return rc;
      }
    }
    if constexpr (md_same<T, double>::value) {
      const bool a_kc = g.a_ks == 1 || g.K == 1, a_mc = g.a_ms == 1 || g.M == 1;
      const bool b_kc = g.b_ks == 1 || g.K == 1, b_nc = g.b_ns == 1 || g.N == 1;
      const int f64_mfma = (int)md_opt(MD_OPT_GEMM_F64_MFMA);
      if (f64_mfma && g.M * g.N >= 64 * 64 && g.K >= 8 && (a_kc || a_mc) && (b_kc || b_nc) && g.M * g.N < (1ll << 40)) {
        GemmArgs64 ga;
        ga.A = (const double *)g.a; ga.B = (const double *)g.b; ga.C = (double *)g.c;
        ga.M = g.M; ga.N = g.N; ga.K = g.K;
        ga.a_bs = g.a_bs; ga.a_ms = g.a_ms; ga.a_ks = g.a_ks;
        ga.b_bs = g.b_bs; ga.b_ks = g.b_ks; ga.b_ns = g.b_ns;
        ga.c_bs = g.c_bs; ga.c_ms = g.c_ms; ga.c_ns = g.c_ns;
        ga.tiles_m = ga.tiles_n = 0;
        const bool A_KC = a_kc && !(a_mc && g.a_ks != 1), B_KC = b_kc && !(b_nc && g.b_ks != 1);
        auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
        bool edge = !al16(g.a) || !al16(g.b);
        edge = edge || ((A_KC ? g.a_ms : g.a_ks) & 1) || ((B_KC ? g.b_ns : g.b_ks) & 1) || (g.a_bs & 1) || (g.b_bs & 1);
        if (A_KC && B_KC) return launch_f64_pick<true, true>(ga, g.batch, edge);
        if (A_KC && !B_KC) return launch_f64_pick<true, false>(ga, g.batch, edge);
        if (!A_KC && B_KC) return launch_f64_pick<false, true>(ga, g.batch, edge);
        return launch_f64_pick<false, false>(ga, g.batch, edge);
      }
    }
    dim3 grid((unsigned)((g.N + 15) / 16), (unsigned)((g.M + 15) / 16), (unsigned)g.batch);
    if (grid.y > 65535) return md_fail(MDHIP_EVALUE, "matmul: M too large for the generic kernel");
    k_gemm_generic<T><<<grid, 256, 0, md_stream()>>>(g);
    return MD_LAUNCH_CHECK("matmul(generic)");
  }
};

}  // namespace

extern "C" int mdhip_matmul(const mdhip_array *a, const mdhip_array *b, const mdhip_array *c) {
  return md_matmul_dispatch<HipExec>(a, b, c);
}

namespace {
template <int BM, int BN, int WM = 2, int WN = 2, bool DMA = false> static int launch_epi(GemmArgs ga) {
  const int64_t tiles = ((ga.M + BM - 1) / BM) * ((ga.N + BN - 1) / BN);
  void *partial = nullptr;
  MD_TRY(mdhip_alloc((size_t)tiles * sizeof(float), &partial));
  ga.sum_out = ga.partial;  // (the caller parked the 0-d result pointer here)
  ga.partial = (float *)partial;
  ga.tickets = md_tickets();
  int rc;
  if constexpr (DMA) {
    rc = launch_kc_glds<BM, BN, WM, WN, false, 1>(ga, 1, false);   // (the caller checked whole tiles and alignment)
    if (rc < 0) rc = md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: shape not covered by the fused kernel");
  }
  else rc = launch_cfg<BM, BN, 16, WM, WN, true, false, BM == 64 ? 0 : 1, 1>(ga, 1, false);
  mdhip_free(partial);
  return rc;
}
}  // namespace

// sum_out = sum(where(a @ b + bias > 0, a @ b + bias, 0)) and mask_out = (a @ b + bias > 0), one GEMM pass with the
// elementwise tail and the reduction in its epilogue. MDHIP_EVALUE ("not fused") for anything but row-major f32
// a (M x K) and b (K x N) with whole aligned tiles: the caller then runs the plain product and the fused tail.
extern "C" int mdhip_matmul_bias_relu_sum(const mdhip_array *a, const mdhip_array *b, const mdhip_array *bias,
                                          const mdhip_array *mask_out, const mdhip_array *sum_out) {
  MD_TRY(md_check_array(a, "matmul a"));
  MD_TRY(md_check_array(b, "matmul b"));
  MD_TRY(md_check_array(bias, "bias"));
  MD_TRY(md_check_array(mask_out, "mask"));
  MD_TRY(md_check_array(sum_out, "sum"));
  if (a->dtype != MDHIP_F32 || b->dtype != MDHIP_F32 || bias->dtype != MDHIP_F32 || sum_out->dtype != MDHIP_F32 || mask_out->dtype != MDHIP_BOOL)
    return md_fail(MDHIP_ETYPE, "matmul_bias_relu_sum: float32 operands, bool mask, float32 sum");
  if (a->ndim != 2 || b->ndim != 2 || mask_out->ndim != 2 || bias->ndim != 1)
    return md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: 2-D operands and mask, 1-D bias");
  const int64_t M = a->shape[0], K = a->shape[1], N = b->shape[1];
  if (b->shape[0] != K || bias->shape[0] != N || mask_out->shape[0] != M || mask_out->shape[1] != N)
    return md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: shapes do not agree");
  const bool row_major = a->strides[1] == 1 && b->strides[1] == 1 && bias->strides[0] == 1 && mask_out->strides[1] == 1 && mask_out->strides[0] == N;
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  // (the epilogue writes the mask in 16-byte vectors and reads the bias per column: an offset bool view or a short bias view
  // passed through the C-ABI must take the general path, not misaligned 16-byte stores)
  if (!row_major || !al16(a->data) || !al16(b->data) || !al16(mask_out->data) || ((uintptr_t)bias->data & 3) || (N & 15) ||
      (a->strides[0] & 3) || (b->strides[0] & 3) || K < 16 || (K % 16))
    return md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: layout not covered by the fused kernel");
  GemmArgs ga{};
  ga.A = (const float *)a->data; ga.B = (const float *)b->data; ga.C = nullptr;
  ga.M = M; ga.N = N; ga.K = K;
  ga.a_ms = a->strides[0]; ga.a_ks = 1; ga.b_ks = b->strides[0]; ga.b_ns = 1;
  ga.c_ms = N; ga.c_ns = 1;
  ga.bias = (const float *)bias->data;
  ga.mask = (uint8_t *)mask_out->data;
  ga.partial = (float *)sum_out->data;
  // the plain kernel's tile choice (and with it the plain product's summation order: a mask recomputed from `a @ b + bias`
  // agrees bit for bit), restricted to the tiles that divide the problem
  const bool dma_ok = md_opt(MD_OPT_GEMM_GLDS) != 0 && K % 32 == 0;
  const int cfg = pick_cfg(ga, 1, false, dma_ok);
  if (dma_ok) {
    if (cfg == CFG_256x256x32 && M % 256 == 0 && N % 256 == 0) return launch_epi<256, 256, 2, 2, true>(ga);
    if (cfg == CFG_128x128_W8 && M % 128 == 0 && N % 128 == 0) return launch_epi<128, 128, 2, 4, true>(ga);
    if ((cfg == CFG_128x128x16 || cfg == CFG_128x128x32) && M % 128 == 0 && N % 128 == 0) return launch_epi<128, 128, 2, 2, true>(ga);
  }
  if ((cfg == CFG_256x128x16 || cfg == CFG_256x256x32) && M % 256 == 0 && N % 128 == 0) return launch_epi<256, 128>(ga);
  if (cfg == CFG_128x128_W8 && M % 128 == 0 && N % 128 == 0) return launch_epi<128, 128, 2, 4>(ga);
  if (M % 128 == 0 && N % 128 == 0 && (cfg == CFG_128x128x16 || cfg == CFG_128x128x32 || cfg == CFG_256x128x16)) return launch_epi<128, 128>(ga);
  if (M % 128 == 0 && N % 64 == 0 && cfg != CFG_64x64x16) return launch_epi<128, 64>(ga);
  if (M % 64 == 0 && N % 64 == 0) return launch_epi<64, 64>(ga);
  return md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: shape not covered by the fused kernel");
}
