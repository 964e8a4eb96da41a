// fusion.hip — one-pass evaluation of recorded elementwise expressions, with an
// optional reduction epilogue (mdhip_vm_eval / mdhip_vm_reduce in include/mdhip.h).
//
// What it replaces: the eager tape launches one kernel per backend call, so the
// backward of a chain such as sum((sin(x)*y)**2) streams every intermediate through
// HBM (100*N bytes for cfg3) and the bias gradient of cfg4 first
// writes g*mask (128 MiB) and then column-sums it. Here the chain is a postfix
// program interpreted per element (csrc/md_vm.h) — the leaves are read once, the
// result is written (or reduced) once: HBM-bound at the *fused* byte count.
//
// Interpreter kernels (all wave-uniform control flow; they serve arrays below the
// run-time-compilation threshold and are the fallback when hiprtc is unavailable —
// arrays of >= 2^18 elements run the specialised kernels of fusion_jit.inc):
//   k_vm_eval_fast     (rows, inner) geometry: leaves contiguous / row- or column-
//                      broadcast / stride-0, 16-B loads+stores
//   k_vm_eval_generic  any <=8-D strides, one element per lane
//   k_vm_reduce_all    full reduction of the program's value (grid-strided sweep,
//                      block partials + finishing block) — fused "...sum()"
//   k_vm_reduce_cols   2-D program reduced over rows: lane owns 4 columns, rows
//                      split over gridDim.y — fused reduce-to-shape (bias gradient)
// Reductions are deterministic (no atomics).
#include <vector>

#include "md_hip.h"
#include "md_vm.h"

extern "C" int mdhip_alloc(size_t, void **);
extern "C" int mdhip_free(void *);

namespace {

constexpr int VG = 1;        // 16-B vector groups per lane per interpreter pass (large arrays take the compiled path)
constexpr int VW = 4 * VG;   // lanes of the operand stack per thread

// leaf load for VG element groups of the (rows, inner) geometry: VG independent 16-B loads
template <class T> struct FastLoader {
  const MdVmDev &P;
  int64_t row[VG], c[VG];
  template <class S> __device__ __forceinline__ void vec(const MdVmLeaf &L, T (&d)[VW]) const {
    MdVec<S, 4> v[VG];
#pragma unroll
    for (int g = 0; g < VG; ++g) v[g] = *reinterpret_cast<const MdVec<S, 4> *>((const S *)L.p + row[g] * L.os + c[g]);
#pragma unroll
    for (int g = 0; g < VG; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (md_same<S, uint8_t>::value) d[4 * g + j] = (T)(v[g].v[j] != 0);
        else d[4 * g + j] = (T)v[g].v[j];
      }
  }
  __device__ __forceinline__ void operator()(int l, T (&d)[VW]) const {
    const MdVmLeaf &L = P.leaf[l];
    if (L.is) {
      switch (L.dtype) {
        case MDHIP_F32: vec<float>(L, d); break;
        case MDHIP_F64: vec<double>(L, d); break;
        case MDHIP_BOOL: vec<uint8_t>(L, d); break;
        case MDHIP_I32: vec<int32_t>(L, d); break;
        default: vec<int64_t>(L, d); break;
      }
    } else {
#pragma unroll
      for (int g = 0; g < VG; ++g) {
        const T s = md_load<T>(L.p, L.dtype, row[g] * L.os);
#pragma unroll
        for (int j = 0; j < 4; ++j) d[4 * g + j] = s;
      }
    }
  }
};
template <class T> struct ScalarLoader {  // one element: leaf offsets precomputed by the caller
  const MdVmDev &P;
  const int64_t *offs;
  __device__ __forceinline__ void operator()(int l, T (&d)[1]) const { d[0] = md_load<T>(P.leaf[l].p, P.leaf[l].dtype, offs[l]); }
};
template <class T> struct FastLoader1 {  // single element on the fast geometry (tails)
  const MdVmDev &P;
  int64_t row, c;
  __device__ __forceinline__ void operator()(int l, T (&d)[1]) const {
    const MdVmLeaf &L = P.leaf[l];
    d[0] = md_load<T>(L.p, L.dtype, row * L.os + (L.is ? c : 0));
  }
};

// Program staging: every block copies the (<= 48 x 16 B) program from the kernarg
// segment into LDS once (static kernarg offsets: the scalar loads overlap), then
// each interpreter step reads its instruction with one broadcast ds_read_b128 that
// was issued one instruction earlier.
__device__ __forceinline__ void vm_stage_program(const MdVmDev &P, MdVmInstr *prog) {
#pragma unroll
  for (int i = 0; i < MDHIP_VM_MAX_INSTR; ++i) {
    if (i < P.n_instr && threadIdx.x == (unsigned)i) prog[i] = P.code[i];
  }
  if (threadIdx.x == 0) { prog[MDHIP_VM_MAX_INSTR].ctrl = 0; prog[MDHIP_VM_MAX_INSTR].imm = 0.0; }
  __syncthreads();
}
struct LdsFetch {
  const MdVmInstr *prog;
  uint4 nxt;
  __device__ __forceinline__ void prefetch(int pc) { nxt = *reinterpret_cast<const uint4 *>(prog + pc); }
  __device__ __forceinline__ void operator()(int, uint32_t &c, double &imm) const {
    c = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt.x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt.z);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt.w);
    imm = __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
  }
};
#define MD_VM_PROLOGUE                                         \
  __shared__ MdVmInstr s_prog[MDHIP_VM_MAX_INSTR + 1];         \
  vm_stage_program(P, s_prog);                                 \
  LdsFetch fetch{s_prog, uint4{0, 0, 0, 0}}

__device__ __forceinline__ void vm_row_col(int64_t v, int64_t nv, int64_t rows, int64_t &row, int64_t &cv) {
  if (rows == 1) { row = 0; cv = v; }
  else if ((uint64_t)v < 0x100000000ull && (uint64_t)nv < 0x100000000ull) {
    uint32_t q = (uint32_t)v / (uint32_t)nv;
    row = q; cv = (uint32_t)v - q * (uint32_t)nv;
  } else { row = v / nv; cv = v - row * nv; }
}

template <class T, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_vm_eval_fast(MdVmDev P, To *out, int64_t rows, int64_t inner) {
  MD_VM_PROLOGUE;
  const int64_t nv = inner >> 2, total = rows * nv;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v0 = gid; v0 < total; v0 += VG * gs) {
    FastLoader<T> ld{P, {}, {}};
    bool ok[VG];
    int64_t orow[VG], ocol[VG];
#pragma unroll
    for (int g = 0; g < VG; ++g) {
      const int64_t v = v0 + g * gs;
      ok[g] = v < total;
      int64_t row, cv;
      vm_row_col(ok[g] ? v : v0, nv, rows, row, cv);   // out-of-range groups recompute group 0, unstored
      ld.row[g] = orow[g] = row;
      ld.c[g] = ocol[g] = cv << 2;
    }
    T r[VW];
    md_vm_run<T, VW>(P.n_instr, fetch, ld, r);
#pragma unroll
    for (int g = 0; g < VG; ++g) {
      if (!ok[g]) continue;
      MdVec<To, 4> o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o.v[j] = md_cast<To>(r[4 * g + j]);
      *reinterpret_cast<MdVec<To, 4> *>(out + orow[g] * inner + ocol[g]) = o;
    }
  }
  if (rows == 1) {
    const int64_t t0 = nv << 2;
    if (gid < inner - t0) {
      FastLoader1<T> ld{P, 0, t0 + gid};
      T r[1];
      md_vm_run<T, 1>(P.n_instr, fetch, ld, r);
      out[t0 + gid] = md_cast<To>(r[0]);
    }
  }
}

// Three / four collapsed axes with a contiguous (or broadcast) inner axis in every leaf — chains under the broadcasts of a
// normalisation layer, `(x * g[:, None, :] + h[None, :, None]) ** 2` — : a lane evaluates one vector of four, the outer position from
// two or three divisions (elementwise.hip's k_ew_axes is the eager counterpart). Interpreter only: was the one-element generic walk.
struct VmAxes {
  int64_t e0, e1, e2, nv, rows;
  int64_t st[MDHIP_VM_MAX_LEAVES][3];
};
template <class T> struct AxesLoader {
  const MdVmDev &P;
  const VmAxes &A;
  int64_t r0, r1, r2, c;
  template <class S> __device__ __forceinline__ void vec(const MdVmLeaf &L, int64_t off, T (&d)[4]) const {
    const MdVec<S, 4> v = *reinterpret_cast<const MdVec<S, 4> *>((const S *)L.p + off + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (md_same<S, uint8_t>::value) d[j] = (T)(v.v[j] != 0);
      else d[j] = (T)v.v[j];
    }
  }
  __device__ __forceinline__ void operator()(int l, T (&d)[4]) const {
    const MdVmLeaf &L = P.leaf[l];
    const int64_t off = r0 * A.st[l][0] + r1 * A.st[l][1] + r2 * A.st[l][2];
    if (L.is) {
      switch (L.dtype) {
        case MDHIP_F32: vec<float>(L, off, d); break;
        case MDHIP_F64: vec<double>(L, off, d); break;
        case MDHIP_BOOL: vec<uint8_t>(L, off, d); break;
        case MDHIP_I32: vec<int32_t>(L, off, d); break;
        default: vec<int64_t>(L, off, d); break;
      }
    } else {
      const T s = md_load<T>(L.p, L.dtype, off);
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = s;
    }
  }
};
template <class T, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_vm_eval_axes(MdVmDev P, VmAxes A, To *out) {
  MD_VM_PROLOGUE;
  const int64_t total = A.rows * A.nv, gs = (int64_t)gridDim.x * blockDim.x;
  const bool narrow = total < (1ll << 31);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gs) {
    int64_t row, r0 = 0, r1, r2;
    if (narrow) {
      const uint32_t w = (uint32_t)v / (uint32_t)A.nv, q = w / (uint32_t)A.e2;
      row = w;
      r2 = w - q * (uint32_t)A.e2;
      r1 = q;
      if (A.e0 != 1) { const uint32_t t = q / (uint32_t)A.e1; r0 = t; r1 = q - t * (uint32_t)A.e1; }
    } else {
      row = v / A.nv;
      const int64_t q = row / A.e2;
      r2 = row - q * A.e2;
      r1 = q;
      if (A.e0 != 1) { r0 = q / A.e1; r1 = q - r0 * A.e1; }
    }
    const int64_t c = (v - row * A.nv) << 2;
    AxesLoader<T> ld{P, A, r0, r1, r2, c};
    T r[4];
    md_vm_run<T, 4>(P.n_instr, fetch, ld, r);
    MdVec<To, 4> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = md_cast<To>(r[j]);
    *reinterpret_cast<MdVec<To, 4> *>(out + row * (A.nv << 2) + c) = o;
  }
}

template <class T, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_vm_eval_generic(MdVmDev P, MdVmIter it, To *out) {
  MD_VM_PROLOGUE;
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += gs) {
    int64_t offs[MDHIP_VM_MAX_LEAVES + 1];
    for (int l = 0; l <= MDHIP_VM_MAX_LEAVES; ++l) offs[l] = 0;
    int64_t lin = i;
    for (int d = it.ndim - 1; d >= 0; --d) {
      const int64_t e = it.shape[d], q = lin / e, r = lin - q * e;
      lin = q;
      for (int l = 0; l < P.n_leaves; ++l) offs[l] += r * it.strides[l][d];
      offs[MDHIP_VM_MAX_LEAVES] += r * it.strides[MDHIP_VM_MAX_LEAVES][d];
    }
    ScalarLoader<T> ld{P, offs};
    T r[1];
    md_vm_run<T, 1>(P.n_instr, fetch, ld, r);
    out[offs[MDHIP_VM_MAX_LEAVES]] = md_cast<To>(r[0]);
  }
}

// ------------------------------------------------------------------ reductions ----
template <class R, class T> __device__ __forceinline__ T vm_block_reduce(T v, T *smem) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = R::combine(v, md_shfl_down(v, d));
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

template <class R, class T>
__global__ void __launch_bounds__(MD_BLOCK) k_vm_reduce_all(MdVmDev P, int64_t rows, int64_t inner, T *partial, unsigned *tickets, T *out) {
  MD_VM_PROLOGUE;
  __shared__ T smem[MD_BLOCK / 64];
  __shared__ unsigned last_flag;
  const int64_t nv = inner >> 2, total = rows * nv;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gs = (int64_t)gridDim.x * blockDim.x;
  T a[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) a[j] = R::template identity<T>();
  for (int64_t v0 = gid; v0 < total; v0 += VG * gs) {
    FastLoader<T> ld{P, {}, {}};
    bool ok[VG];
#pragma unroll
    for (int g = 0; g < VG; ++g) {
      const int64_t v = v0 + g * gs;
      ok[g] = v < total;
      int64_t row, cv;
      vm_row_col(ok[g] ? v : v0, nv, rows, row, cv);
      ld.row[g] = row;
      ld.c[g] = cv << 2;
    }
    T r[VW];
    md_vm_run<T, VW>(P.n_instr, fetch, ld, r);
#pragma unroll
    for (int g = 0; g < VG; ++g)
      if (ok[g]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[4 * g + j] = R::combine(a[4 * g + j], r[4 * g + j]);
      }
  }
  T acc = R::template identity<T>();
#pragma unroll
  for (int j = 0; j < VW; ++j) acc = R::combine(acc, a[j]);
  if (rows == 1) {
    const int64_t t0 = nv << 2;
    if (gid < inner - t0) {
      FastLoader1<T> ld{P, 0, t0 + gid};
      T r[1];
      md_vm_run<T, 1>(P.n_instr, fetch, ld, r);
      acc = R::combine(acc, r[0]);
    }
  }
  acc = vm_block_reduce<R>(acc, smem);
  // block partials -> the block that arrives last sums them in index order (md_ticket.h): no finishing launch
  if (threadIdx.x == 0) md_st_sc1(partial + blockIdx.x, acc);
  const bool last = gridDim.x >= 64 ? md_ticket_last2(tickets, blockIdx.x, gridDim.x, &last_flag) : md_ticket_last(tickets, gridDim.x, &last_flag);
  if (!last) return;
  acc = md_fold_partials<R>(partial, gridDim.x);
  acc = vm_block_reduce<R>(acc, smem);
  if (threadIdx.x == 0) out[0] = acc;
}

// 2-D program [n_red rows][n_out cols] reduced over rows; a lane owns 4 columns and
// takes rows r, r+4, r+8, r+12 per interpreter pass (four 16-B loads per leaf in flight).
template <class R, class T, bool FINAL>
__global__ void __launch_bounds__(MD_BLOCK) k_vm_reduce_cols(MdVmDev P, int64_t n_out, int64_t n_red, int64_t chunk, T *dst) {
  MD_VM_PROLOGUE;
  __shared__ T smem[3][64][4];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t col = ((int64_t)blockIdx.x * 64 + cx) * 4;
  const int64_t s = blockIdx.y, r0 = s * chunk;
  int64_t r1 = r0 + chunk;
  if (r1 > n_red) r1 = n_red;
  T a[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) a[j] = R::template identity<T>();
  // every lane of a wave runs the interpreter together (uniform control flow); lanes past
  // the last column recompute column group 0 of their block and are never stored
  const int64_t lcol = col < n_out ? col : (int64_t)blockIdx.x * 256;
  for (int64_t r = r0 + ry; r < r1; r += 4 * VG) {
    FastLoader<T> ld{P, {}, {}};
    bool ok[VG];
#pragma unroll
    for (int g = 0; g < VG; ++g) {
      const int64_t rr = r + 4 * g;
      ok[g] = rr < r1;
      ld.row[g] = ok[g] ? rr : r;
      ld.c[g] = lcol;
    }
    T v[VW];
    md_vm_run<T, VW>(P.n_instr, fetch, ld, v);
#pragma unroll
    for (int g = 0; g < VG; ++g)
      if (ok[g]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[4 * g + j] = R::combine(a[4 * g + j], v[4 * g + j]);
      }
  }
  T tot[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    tot[j] = a[j];
#pragma unroll
    for (int g = 1; g < VG; ++g) tot[j] = R::combine(tot[j], a[4 * g + j]);
  }
  if (ry > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) smem[ry - 1][cx][j] = tot[j];
  }
  __syncthreads();
  if (ry == 0 && col < n_out) {
    MdVec<T, 4> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = R::combine(R::combine(tot[j], smem[0][cx][j]), R::combine(smem[1][cx][j], smem[2][cx][j]));
    T *d = FINAL ? dst + col : dst + s * n_out + col;
    *reinterpret_cast<MdVec<T, 4> *>(d) = o;
  }
}

}  // namespace
#include "fusion_jit.inc"
namespace {

// ------------------------------------------------------------------------ host ----
static void to_dev(const mdhip_vm_program *pr, MdVmDev *D) {
  memset(D, 0, sizeof *D);
  D->n_instr = pr->n_instr;
  D->n_leaves = pr->n_leaves;
  for (int i = 0; i < pr->n_instr; ++i) { D->code[i].ctrl = pr->ctrl[i]; D->code[i].imm = pr->imm[i]; }
  for (int l = 0; l < pr->n_leaves; ++l) {
    D->leaf[l].p = pr->leaves[l].data;
    D->leaf[l].dtype = pr->leaves[l].dtype;
  }
}
static bool al_for(const void *p, int dtype) {
  const uintptr_t al = md_dtype_size(dtype) * 4 > 16 ? 16 : md_dtype_size(dtype) * 4;
  return ((uintptr_t)p % al) == 0;
}
// (rows, inner) geometry usable by the vector kernels? fills leaf os/is.
static bool fast_geometry(const MdVmIter &it, int n_leaves, const mdhip_vm_program *pr, bool out_contig_required, MdVmDev *D,
                          int64_t *rows, int64_t *inner) {
  if (it.ndim < 1 || it.ndim > 2) return false;
  *inner = it.shape[it.ndim - 1];
  *rows = it.ndim == 2 ? it.shape[0] : 1;
  if (*rows > 1 && (*inner & 3)) return false;
  const int O = MDHIP_VM_MAX_LEAVES;
  if (out_contig_required) {
    if (it.strides[O][it.ndim - 1] != 1) return false;
    if (it.ndim == 2 && it.strides[O][0] != *inner) return false;
  }
  for (int l = 0; l < n_leaves; ++l) {
    const int64_t is = it.strides[l][it.ndim - 1], os = it.ndim == 2 ? it.strides[l][0] : 0;
    if (is != 0 && is != 1) return false;
    if (is == 1) {
      if (!al_for(pr->leaves[l].data, pr->leaves[l].dtype)) return false;
      if (*rows > 1 && (os & 3)) return false;
    }
    D->leaf[l].os = os;
    D->leaf[l].is = (int32_t)is;
  }
  return true;
}

// k_vm_eval_axes eligibility (the conditions of elementwise.hip's axes_geom, per leaf); fills leaf `is` and the stride table
static bool axes_geometry(const MdVmIter &it, const mdhip_vm_program *pr, const mdhip_array *out, MdVmDev *D, VmAxes *A) {
  if (it.ndim != 3 && it.ndim != 4) return false;
  const int nd = it.ndim, sh = 4 - nd, O = MDHIP_VM_MAX_LEAVES;
  const int64_t inner = it.shape[nd - 1];
  if ((inner & 3) || it.total < (1 << 16) || it.strides[O][nd - 1] != 1 || !al_for(out->data, out->dtype)) return false;
  int64_t dense = inner;
  for (int d = nd - 2; d >= 0; --d) {
    if (it.strides[O][d] != dense) return false;
    dense *= it.shape[d];
  }
  memset(A, 0, sizeof *A);
  A->e0 = nd == 4 ? it.shape[0] : 1;
  A->e1 = it.shape[1 - sh];
  A->e2 = it.shape[2 - sh];
  A->nv = inner >> 2;
  A->rows = A->e0 * A->e1 * A->e2;
  for (int l = 0; l < pr->n_leaves; ++l) {
    const int64_t is = it.strides[l][nd - 1];
    if (is != 0 && is != 1) return false;
    if (is == 1 && !al_for(pr->leaves[l].data, pr->leaves[l].dtype)) return false;
    for (int j = sh; j < 3; ++j) {
      const int64_t st = it.strides[l][j - sh];
      if (is == 1 && (st & 3)) return false;
      A->st[l][j] = st;
    }
    D->leaf[l].os = 0;
    D->leaf[l].is = (int32_t)is;
  }
  return true;
}

template <class T> static int eval_typed(const mdhip_vm_program *pr, const mdhip_array *out) {
  MdVmIter it;
  MD_TRY(md_vm_build_iter(&it, pr, out, out));
  if (it.total == 0) return MDHIP_OK;
  MdVmDev D;
  to_dev(pr, &D);
  hipStream_t st = md_stream();
  int64_t rows, inner;
  const bool to_bool = out->dtype == MDHIP_BOOL;
  if (fast_geometry(it, pr->n_leaves, pr, true, &D, &rows, &inner) && al_for(out->data, out->dtype)) {
    if (jit::enabled() && it.total >= jit::min_elems()) {
      // the streamed bytes of this launch: output + distinct vector leaves
      int64_t bytes = it.total * (int64_t)md_dtype_size(out->dtype);
      for (int l = 0; l < pr->n_leaves; ++l)
        if (D.leaf[l].is) bytes += it.total * (int64_t)md_dtype_size(pr->leaves[l].dtype);
      jit::Spec S;
      S.kind = jit::EVAL;
      jit::spec_single(&S, pr);
      jit::spec_modes(&S, D, rows);
      S.out_bool = to_bool;
      S.nt = bytes > ((int64_t)320 << 20);
      jit::eval_shape(&S);
      if (hipFunction_t fn = jit::get(S)) {
        jit::JArgs A;
        jit::fill_args(&A, pr, D);
        A.outs[0] = out->data;
        const int grid = jit::stream_grid(&A, rows, inner, jit::eval_blocks(S));
        return jit::launch(fn, A, dim3((unsigned)grid));
      }
    }
    const int64_t work = (rows * (inner >> 2) + VG - 1) / VG + (rows == 1 ? 4 : 0);
    if (to_bool) k_vm_eval_fast<T, b8><<<md_grid_for(work), MD_BLOCK, 0, st>>>(D, (b8 *)out->data, rows, inner);
    else k_vm_eval_fast<T, T><<<md_grid_for(work), MD_BLOCK, 0, st>>>(D, (T *)out->data, rows, inner);
    return MD_LAUNCH_CHECK("vm_eval(fast)");
  }
  VmAxes A;
  if (axes_geometry(it, pr, out, &D, &A)) {
    if (to_bool) k_vm_eval_axes<T, b8><<<md_grid_for(it.total >> 2), MD_BLOCK, 0, st>>>(D, A, (b8 *)out->data);
    else k_vm_eval_axes<T, T><<<md_grid_for(it.total >> 2), MD_BLOCK, 0, st>>>(D, A, (T *)out->data);
    return MD_LAUNCH_CHECK("vm_eval(axes)");
  }
  if (to_bool) k_vm_eval_generic<T, b8><<<md_grid_for(it.total), MD_BLOCK, 0, st>>>(D, it, (b8 *)out->data);
  else k_vm_eval_generic<T, T><<<md_grid_for(it.total), MD_BLOCK, 0, st>>>(D, it, (T *)out->data);
  return MD_LAUNCH_CHECK("vm_eval(generic)");
}

// merge the leaf tables of n programs (identical descriptors share a slot) and give every
// immediate its slot in the shared array; false if the bounds of one launch are exceeded
static bool merge_programs(const mdhip_vm_program *progs, int n, jit::Spec *M, mdhip_vm_program *merged) {
  if (n < 2 || n > 4) return false;
  M->kind = jit::EVAL;
  M->n = n;
  M->n_leaves = 0;
  M->n_imm = 0;
  M->f32 = progs[0].compute_dtype == MDHIP_F32;
  memset(merged, 0, sizeof *merged);
  merged->compute_dtype = progs[0].compute_dtype;
  for (int k = 0; k < n; ++k) {
    const mdhip_vm_program *pr = &progs[k];
    if (pr->compute_dtype != progs[0].compute_dtype) return false;
    M->pr[k] = pr;
    for (int l = 0; l < pr->n_leaves; ++l) {
      const mdhip_array &a = pr->leaves[l];
      int slot = -1;
      for (int m = 0; m < M->n_leaves && slot < 0; ++m) {
        const mdhip_array &b = merged->leaves[m];
        bool same = a.data == b.data && a.dtype == b.dtype && a.ndim == b.ndim;
        for (int d = 0; same && d < a.ndim; ++d) same = a.shape[d] == b.shape[d] && a.strides[d] == b.strides[d];
        if (same) slot = m;
      }
      if (slot < 0) {
        if (M->n_leaves == MDHIP_VM_MAX_LEAVES) return false;
        slot = M->n_leaves++;
        merged->leaves[slot] = a;
        M->leaf_dtype[slot] = a.dtype;
      }
      M->leaf_map[k][l] = slot;
    }
    for (int pc = 0; pc < pr->n_instr; ++pc) {
      const uint32_t c = pr->ctrl[pc];
      const bool has_imm = MD_VM_KIND(c) != MDHIP_VM_UNARY && MD_VM_KIND(c) != MDHIP_VM_WHERE &&
                           (MD_VM_RS(c) == MDHIP_VM_SRC_CONST || (MD_VM_KIND(c) == MDHIP_VM_BINARY && MD_VM_LS(c) == MDHIP_VM_SRC_CONST));
      M->imm_slot[k][pc] = 0;
      if (has_imm) {
        if (M->n_imm == MDHIP_VM_MAX_INSTR) return false;
        merged->imm[M->n_imm] = pr->imm[pc];
        M->imm_slot[k][pc] = M->n_imm++;
      }
    }
  }
  merged->n_leaves = M->n_leaves;
  merged->n_instr = 1;  // a placeholder PUSH so that the merged table passes md_vm_build_iter
  merged->ctrl[0] = MDHIP_VM_CTRL(MDHIP_VM_PUSH, 0, 0, 0, MDHIP_VM_SRC_LEAF, 0);
  return M->n_leaves > 0;
}

static int eval_multi(const mdhip_vm_program *progs, const mdhip_array *outs, int n, bool *done) {
  *done = false;
  if (!jit::enabled()) return MDHIP_OK;
  jit::Spec M;
  mdhip_vm_program merged;
  if (!merge_programs(progs, n, &M, &merged)) return MDHIP_OK;
  for (int k = 0; k < n; ++k) {  // outputs: the compute dtype, contiguous, one shape
    if (outs[k].dtype != merged.compute_dtype || outs[k].ndim != outs[0].ndim) return MDHIP_OK;
    for (int d = 0; d < outs[0].ndim; ++d)
      if (outs[k].shape[d] != outs[0].shape[d] || outs[k].strides[d] != outs[0].strides[d]) return MDHIP_OK;
    if (((uintptr_t)outs[k].data & 15) != 0) return MDHIP_OK;
  }
  MdVmIter it;
  if (md_vm_build_iter(&it, &merged, &outs[0], &outs[0]) != MDHIP_OK) return MDHIP_OK;
  if (it.total < jit::min_elems()) return MDHIP_OK;
  MdVmDev D;
  to_dev(&merged, &D);
  int64_t rows, inner;
  if (!fast_geometry(it, merged.n_leaves, &merged, true, &D, &rows, &inner)) return MDHIP_OK;
  int64_t bytes = (int64_t)n * it.total * (int64_t)md_dtype_size(merged.compute_dtype);
  for (int l = 0; l < merged.n_leaves; ++l)
    if (D.leaf[l].is) bytes += it.total * (int64_t)md_dtype_size(merged.leaves[l].dtype);
  jit::spec_modes(&M, D, rows);
  M.nt = bytes > ((int64_t)320 << 20);
  jit::eval_shape(&M);
  hipFunction_t fn = jit::get(M);
  if (!fn) return MDHIP_OK;
  jit::JArgs A;
  memset(&A, 0, sizeof A);
  for (int l = 0; l < merged.n_leaves; ++l) { A.leaf[l].p = D.leaf[l].p; A.leaf[l].os = D.leaf[l].os; A.leaf[l].is = D.leaf[l].is; }
  for (int i = 0; i < M.n_imm; ++i) A.imm[i] = merged.imm[i];
  for (int k = 0; k < n; ++k) A.outs[k] = outs[k].data;
  const int grid = jit::stream_grid(&A, rows, inner, jit::eval_blocks(M));
  *done = true;
  return jit::launch(fn, A, dim3((unsigned)grid));
}

static int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A program whose operands are all dense or fully broadcast collapses to ONE axis; the column reductions want it as
// (R rows, C columns) again: dense leaves advance C elements per row, broadcast ones none.
static bool uncollapse_2d(const MdVmIter &it, int n_leaves, MdVmDev *D, int64_t R, int64_t C, int64_t *rows, int64_t *inner) {
  if (it.ndim == 2 && *rows == R && *inner == C) return true;
  if (it.ndim != 1 || *rows != 1 || *inner != R * C || (C & 3)) return false;
  for (int l = 0; l < n_leaves; ++l) D->leaf[l].os = D->leaf[l].is ? C : 0;
  *rows = R;
  *inner = C;
  return true;
}

// Strip geometry for a (n_red x n_out) program (the generated SWEEP kernel, fusion_jit.inc): NS strips of 256 columns x NB
// interleaved row bands, one block per CU, <= 64 bands (the last block of a strip holds its partial rows in registers).
static bool sweep_geometry(int64_t n_out, int64_t n_red, int ru, int64_t *NS, int64_t *NB) {
  if ((n_out & 3) || n_out < 4 || n_red < 512) return false;
  const int nb_force = (int)md_opt(MD_OPT_SWEEP_NB);
  const int64_t ns = (n_out + 255) / 256;
  int64_t nb = nb_force > 0 ? nb_force : (ns >= MD_NUM_CUS ? 1 : MD_NUM_CUS / ns);
  if (nb > 64) nb = 64;
  if (nb > n_red / (4 * ru)) nb = n_red / (4 * ru);
  if (nb < 1) nb = 1;
  if (ns * nb >= (1ll << 31) || (nb > 1 && ns * MD_TICKET_PAD > MD_TICKET_WORDS)) return false;
  *NS = ns;
  *NB = nb;
  return true;
}

// reduce over axis 0 of a 2-D program with the generated sweep kernel; `eval_out` != nullptr: the evaluated value
// is written as well (one pass for an elementwise product and its reduce-to-shape). false: not applicable.
template <class R, class T>
static bool sweep_cols(const mdhip_vm_program *pr, int rop, const MdVmDev &D, int64_t rows, int64_t inner, void *eval_out,
                       const mdhip_array *out, int *status) {
  if (!jit::enabled() || rows * inner < jit::min_elems()) return false;
  jit::Spec S;
  S.kind = jit::SWEEP;
  jit::spec_single(&S, pr);
  jit::spec_modes(&S, D, rows);
  S.rop = rop;
  S.store = eval_out != nullptr;
  int64_t bytes = eval_out ? rows * inner * (int64_t)sizeof(T) : 0;
  for (int l = 0; l < pr->n_leaves; ++l)
    if (S.leaf_mode[l] == jit::LM_VEC) bytes += rows * inner * (int64_t)md_dtype_size(pr->leaves[l].dtype);
  // rows per trip: ~128 B of loads in flight per lane (a bool leaf brings 4 B per row, a float leaf 16)
  int64_t row_bytes = 0;
  for (int l = 0; l < pr->n_leaves; ++l)
    if (S.leaf_mode[l] == jit::LM_VEC) row_bytes += 4 * (int64_t)md_dtype_size(pr->leaves[l].dtype);
  const int ru_env = (int)md_opt(MD_OPT_SWEEP_RU);
  S.RU = ru_env > 0 ? ru_env : (row_bytes > 0 ? (int)((128 + row_bytes - 1) / row_bytes) : 1);
  if (S.RU > 8) S.RU = 8;
  if (S.RU < 1) S.RU = 1;
  int64_t NS, NB;
  if (!sweep_geometry(inner, rows, S.RU, &NS, &NB)) return false;
  S.nt = bytes > ((int64_t)320 << 20);
  // the evaluated value of a one-pass eval + column reduce is a large write next to (often much smaller) reads, and its
  // reader is whatever needed it in memory (cfg4: the weight-gradient GEMM, which is not bandwidth-bound): written around
  // the caches the pass ran 32.9 -> 26.3 us on the cfg4 shape (128 MiB out, 32 MiB mask in), the GEMM behind it unchanged (option sweep_nt_store = 0 disables)
  const int nt_store = (int)md_opt(MD_OPT_SWEEP_NT_STORE);
  S.nt_store = nt_store && eval_out != nullptr && rows * inner * (int64_t)sizeof(T) >= ((int64_t)64 << 20);
  hipFunction_t fn = jit::get(S);
  if (!fn) return false;
  void *partial = nullptr;
  if (NB > 1) {
    *status = mdhip_alloc((size_t)(NB * inner) * sizeof(T), &partial);
    if (*status != MDHIP_OK) return true;
  }
  jit::JArgs A;
  jit::fill_args(&A, pr, D);
  A.rows = rows; A.inner = inner; A.n_out = inner; A.n_red = rows; A.chunk = NB;
  A.out = out->data;
  A.outs[0] = eval_out;
  A.partial = partial;
  A.tickets = md_tickets();
  *status = jit::launch(fn, A, dim3((unsigned)(NS * NB)));
  if (partial) mdhip_free(partial);  // stream-ordered
  return true;
}

template <class R, class T>
static int reduce_typed(const mdhip_vm_program *pr, int rop, const mdhip_array *shape_like, const mdhip_array *out, uint32_t mask) {
  const int nd = shape_like->ndim;
  const uint32_t all = nd ? ((1u << nd) - 1u) : 0u;
  MdVmIter it;
  MD_TRY(md_vm_build_iter(&it, pr, shape_like, nullptr));
  if (it.total == 0) return md_fail(MDHIP_EVALUE, "vm_reduce: empty operand");
  MdVmDev D;
  to_dev(pr, &D);
  hipStream_t st = md_stream();
  int64_t rows, inner;
  if (!fast_geometry(it, pr->n_leaves, pr, false, &D, &rows, &inner)) return md_fail(MDHIP_EVALUE, "vm_reduce: geometry not supported");
  if (((uintptr_t)out->data & 15) != 0) return md_fail(MDHIP_EVALUE, "vm_reduce: unaligned output");
  if (mask == all) {
    int64_t rbytes = 0;
    for (int l = 0; l < pr->n_leaves; ++l)
      if (D.leaf[l].is) rbytes += it.total * (int64_t)md_dtype_size(pr->leaves[l].dtype);
    hipFunction_t fn = nullptr;
    if (jit::enabled() && it.total >= jit::min_elems()) {
      jit::Spec S;
      S.kind = jit::RED_ALL;
      jit::spec_single(&S, pr);
      jit::spec_modes(&S, D, rows);
      S.rop = rop;
      S.nt = rbytes > ((int64_t)320 << 20);
      fn = jit::get(S);
    }
    jit::JArgs A;
    int grid;
    if (fn) {
      jit::fill_args(&A, pr, D);
      grid = jit::stream_grid(&A, rows, inner, 4);   // read-only streams: 4 blocks per CU (cfg3 loss pass 137-143 -> 126 us)
    } else {
      grid = md_grid_for((rows * (inner >> 2) + VG - 1) / VG + (rows == 1 ? 4 : 0));
    }
    void *partial = nullptr;
    MD_TRY(mdhip_alloc((size_t)grid * sizeof(T), &partial));
    int rc;
    if (fn) {
      A.out = out->data;
      A.partial = partial;
      A.tickets = md_tickets();
      rc = jit::launch(fn, A, dim3((unsigned)grid));
    } else {
      k_vm_reduce_all<R, T><<<grid, MD_BLOCK, 0, st>>>(D, rows, inner, (T *)partial, md_tickets(), (T *)out->data);
      rc = MD_LAUNCH_CHECK("vm_reduce(all)");
    }
    mdhip_free(partial);
    return rc;
  }
  // reduce over axis 0 of a 2-D program that did NOT collapse to 1-D
  if (nd == 2 && mask == 1u && uncollapse_2d(it, pr->n_leaves, &D, shape_like->shape[0], shape_like->shape[1], &rows, &inner)) {
    int status = MDHIP_OK;
    if (sweep_cols<R, T>(pr, rop, D, rows, inner, nullptr, out, &status)) return status;
    const int64_t n_out = inner, n_red = rows;
    const int64_t bx = ceil_div(n_out, 256);
    int64_t splits = 1024 / bx;
    if (splits > n_red / 32) splits = n_red / 32;
    if (splits > 65535) splits = 65535;
    if (splits < 1) splits = 1;
    const int64_t chunk = ceil_div(ceil_div(n_red, splits), 16) * 16;
    splits = ceil_div(n_red, chunk);
    hipFunction_t fn = nullptr;
    if (jit::enabled() && it.total >= jit::min_elems()) {
      jit::Spec S;
      S.kind = jit::RED_COLS;
      jit::spec_single(&S, pr);
      jit::spec_modes(&S, D, rows);
      S.rop = rop;
      fn = jit::get(S);
    }
    jit::JArgs A;
    if (fn) {
      jit::fill_args(&A, pr, D);
      A.rows = rows; A.inner = inner; A.n_out = n_out; A.n_red = n_red; A.chunk = chunk;
    }
    if (splits == 1) {
      if (fn) { A.out = out->data; return jit::launch(fn, A, dim3((unsigned)bx, 1)); }
      k_vm_reduce_cols<R, T, true><<<dim3((unsigned)bx, 1), MD_BLOCK, 0, st>>>(D, n_out, n_red, chunk, (T *)out->data);
      return MD_LAUNCH_CHECK("vm_reduce(cols)");
    }
    void *partial = nullptr;
    MD_TRY(mdhip_alloc((size_t)(splits * n_out) * sizeof(T), &partial));
    if (fn) {
      A.out = partial;
      int rc = jit::launch(fn, A, dim3((unsigned)bx, (unsigned)splits));
      if (rc != MDHIP_OK) { mdhip_free(partial); return rc; }
    } else {
      k_vm_reduce_cols<R, T, false><<<dim3((unsigned)bx, (unsigned)splits), MD_BLOCK, 0, st>>>(D, n_out, n_red, chunk, (T *)partial);
    }
    // second pass: a one-instruction program over the partial buffer
    MdVmDev F;
    memset(&F, 0, sizeof F);
    F.n_instr = 1; F.n_leaves = 1;
    F.code[0].ctrl = MDHIP_VM_CTRL(MDHIP_VM_PUSH, 0, 0, 0, MDHIP_VM_SRC_LEAF, 0);
    F.leaf[0].p = partial; F.leaf[0].os = n_out; F.leaf[0].is = 1; F.leaf[0].dtype = md_dtype_of<T>::value;
    const int64_t chunk2 = ceil_div(splits, 16) * 16;
    k_vm_reduce_cols<R, T, true><<<dim3((unsigned)bx, 1), MD_BLOCK, 0, st>>>(F, n_out, splits, chunk2, (T *)out->data);
    int rc = MD_LAUNCH_CHECK("vm_reduce(cols,split)");
    mdhip_free(partial);
    return rc;
  }
  return md_fail(MDHIP_EVALUE, "vm_reduce: only full reductions and axis-0 reductions of 2-D programs are fused");
}

// out_eval[r][c] = program(r, c) and out_red[c] = reduce over r, ONE pass (generated sweep kernel only)
template <class R, class T>
static int eval_reduce_cols_typed(const mdhip_vm_program *pr, int rop, const mdhip_array *out_eval, const mdhip_array *out_red) {
  MdVmIter it;
  MD_TRY(md_vm_build_iter(&it, pr, out_eval, out_eval));
  if (it.total == 0 || out_eval->ndim != 2) return md_fail(MDHIP_EVALUE, "vm_eval_reduce_cols: a non-empty 2-D program is required");
  MdVmDev D;
  to_dev(pr, &D);
  int64_t rows, inner;
  if (!fast_geometry(it, pr->n_leaves, pr, true, &D, &rows, &inner) ||
      !uncollapse_2d(it, pr->n_leaves, &D, out_eval->shape[0], out_eval->shape[1], &rows, &inner))
    return md_fail(MDHIP_EVALUE, "vm_eval_reduce_cols: geometry not supported");
  if (((uintptr_t)out_eval->data & 15) || ((uintptr_t)out_red->data & 15)) return md_fail(MDHIP_EVALUE, "vm_eval_reduce_cols: unaligned output");
  int status = MDHIP_OK;
  if (sweep_cols<R, T>(pr, rop, D, rows, inner, out_eval->data, out_red, &status)) return status;
  return md_fail(MDHIP_EVALUE, "vm_eval_reduce_cols: shape not covered by the one-pass kernel");
}

// compile-only probes have no launch geometry: read modes from the leaf descriptors as given
static void probe_modes(jit::Spec *S, const mdhip_vm_program *pr) {
  for (int l = 0; l < pr->n_leaves; ++l) {
    const mdhip_array &a = pr->leaves[l];
    bool all0 = true;
    for (int d = 0; d < a.ndim; ++d) all0 = all0 && (a.strides[d] == 0 || a.shape[d] == 1);
    const bool inner0 = a.ndim == 0 || a.strides[a.ndim - 1] == 0;
    const int slot = S->leaf_map[0][l];
    if (all0) S->leaf_mode[slot] = jit::LM_CONST;
    else if (inner0) S->leaf_mode[slot] = jit::LM_ROWB;
    else if (S->kind == jit::SWEEP && a.ndim == 2 && a.strides[0] == 0) S->leaf_mode[slot] = jit::LM_ROWINV;
    else S->leaf_mode[slot] = jit::LM_VEC;
  }
}

}  // namespace

extern "C" {

int mdhip_vm_eval(const mdhip_vm_program *pr, const mdhip_array *out) {
  MD_TRY(md_vm_check(pr));
  MD_TRY(md_check_array(out, "vm out"));
  if (out->dtype != pr->compute_dtype && out->dtype != MDHIP_BOOL)
    return md_fail(MDHIP_ETYPE, "vm_eval: out dtype must be the compute dtype or bool");
  return pr->compute_dtype == MDHIP_F32 ? eval_typed<float>(pr, out) : eval_typed<double>(pr, out);
}

int mdhip_vm_eval_multi(const mdhip_vm_program *progs, const mdhip_array *outs, int n) {
  if (n < 1) return md_fail(MDHIP_EVALUE, "vm_eval_multi: no programs");
  for (int k = 0; k < n; ++k) {
    MD_TRY(md_vm_check(&progs[k]));
    MD_TRY(md_check_array(&outs[k], "vm out"));
  }
  bool done = false;
  MD_TRY(eval_multi(progs, outs, n, &done));
  if (done) return MDHIP_OK;
  for (int k = 0; k < n; ++k) MD_TRY(mdhip_vm_eval(&progs[k], &outs[k]));  // same results, one pass each
  return MDHIP_OK;
}

int mdhip_vm_jit_probe_multi(const mdhip_vm_program *progs, int n, char *log, size_t log_cap) {
  for (int k = 0; k < n; ++k) MD_TRY(md_vm_check(&progs[k]));
  jit::Spec M;
  mdhip_vm_program merged;
  if (!merge_programs(progs, n, &M, &merged)) return md_fail(MDHIP_EVALUE, "jit probe: programs cannot share one launch");
  for (int k = 0; k < n; ++k) {
    jit::Spec one;
    one.kind = jit::EVAL;
    for (int l = 0; l < progs[k].n_leaves; ++l) one.leaf_map[0][l] = M.leaf_map[k][l];
    probe_modes(&one, &progs[k]);
    for (int l = 0; l < progs[k].n_leaves; ++l) M.leaf_mode[M.leaf_map[k][l]] = one.leaf_mode[M.leaf_map[k][l]];
  }
  std::vector<char> code;
  std::string l;
  const int rc = jit::compile(jit::gen_source(M, "k_fused_probe"), &code, &l);
  if (log && log_cap) { strncpy(log, l.c_str(), log_cap - 1); log[log_cap - 1] = 0; }
  if (rc != 0) return md_fail(MDHIP_ERUNTIME, "fused-kernel compilation failed: %.300s", l.c_str());
  return MDHIP_OK;
}

int mdhip_vm_jit_probe(const mdhip_vm_program *pr, int kind, int reduce_op, int out_is_bool, char *log, size_t log_cap) {
  MD_TRY(md_vm_check(pr));
  if (kind < 0 || kind > 4)
    return md_fail(MDHIP_EVALUE, "jit probe: kind must be 0 (eval), 1 (reduce all), 2 (reduce columns, tiled), 3 (reduce columns, sweep) or 4 (eval + reduce columns)");
  jit::Spec S;
  S.kind = kind >= 3 ? jit::SWEEP : kind;
  jit::spec_single(&S, pr);
  probe_modes(&S, pr);
  S.rop = reduce_op;
  S.out_bool = out_is_bool != 0 && kind == 0;
  S.store = kind == 4;
  S.Q = 4;
  S.RU = 2;
  std::vector<char> code;
  std::string l;
  const int rc = jit::compile(jit::gen_source(S, "k_fused_probe"), &code, &l);
  if (log && log_cap) { strncpy(log, l.c_str(), log_cap - 1); log[log_cap - 1] = 0; }
  if (rc != 0) return md_fail(MDHIP_ERUNTIME, "fused-kernel compilation failed: %.300s", l.c_str());
  return MDHIP_OK;
}

int mdhip_vm_eval_reduce_cols(const mdhip_vm_program *pr, int op, const mdhip_array *out_eval, const mdhip_array *out_red) {
  MD_TRY(md_vm_check(pr));
  MD_TRY(md_check_array(out_eval, "vm out"));
  MD_TRY(md_check_array(out_red, "vm out"));
  if (out_eval->dtype != pr->compute_dtype || out_red->dtype != pr->compute_dtype)
    return md_fail(MDHIP_ETYPE, "vm_eval_reduce_cols: both outputs must have the compute dtype");
#define MD_VMER(R)                                                                                        \
  return pr->compute_dtype == MDHIP_F32 ? eval_reduce_cols_typed<R, float>(pr, op, out_eval, out_red)     \
                                        : eval_reduce_cols_typed<R, double>(pr, op, out_eval, out_red)
  switch (op) {
    case MDHIP_R_SUM: MD_VMER(RSum);
    case MDHIP_R_PROD: MD_VMER(RProd);
    case MDHIP_R_MAX: MD_VMER(RMax);
    case MDHIP_R_MIN: MD_VMER(RMin);
  }
#undef MD_VMER
  return md_fail(MDHIP_EVALUE, "vm_eval_reduce_cols: reduce op %d is not fused", op);
}

int mdhip_vm_jit_stats(int64_t stats[2]) {
  stats[0] = jit::g_compiled;
  stats[1] = jit::g_launched;
  return MDHIP_OK;
}

int mdhip_vm_reduce(const mdhip_vm_program *pr, int op, const mdhip_array *shape_like, const mdhip_array *out, uint32_t mask) {
  MD_TRY(md_vm_check(pr));
  MD_TRY(md_check_array(shape_like, "vm shape"));
  MD_TRY(md_check_array(out, "vm out"));
  if (out->dtype != pr->compute_dtype) return md_fail(MDHIP_ETYPE, "vm_reduce: out dtype must be the compute dtype");
#define MD_VMR(R)                                                                                         \
  return pr->compute_dtype == MDHIP_F32 ? reduce_typed<R, float>(pr, op, shape_like, out, mask)           \
                                        : reduce_typed<R, double>(pr, op, shape_like, out, mask)
  switch (op) {
    case MDHIP_R_SUM: MD_VMR(RSum);
    case MDHIP_R_PROD: MD_VMR(RProd);
    case MDHIP_R_MAX: MD_VMR(RMax);
    case MDHIP_R_MIN: MD_VMR(RMin);
  }
#undef MD_VMR
  return md_fail(MDHIP_EVALUE, "vm_reduce: reduce op %d is not fused", op);
}

}  // extern "C"
