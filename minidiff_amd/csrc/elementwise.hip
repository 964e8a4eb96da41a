// elementwise.hip — unary / binary / where / fill / arange kernels for gfx950.
//
// Serves the backend names of reference minidiff/backend/numpy.py:19-95 (unary
// table, binary table, clip/where) and the creation family :98-103,125. All of
// them are HBM-bound: one pass, every input byte read once,
// every output byte written once.
//
// Two kernel families per arity:
//  * fast  — output contiguous, iteration space collapsed to (rows, inner) with
//            every operand's inner stride 0 or 1 (contiguous, row-broadcast such
//            as the bias add, column-broadcast, stride-0 views of a 0-d seed, or a
//            host scalar). 4 elements per lane: 16-B loads/stores for f32/i32,
//            2x16 B for f64/i64, 4 B for bool masks; grid capped at 8 blocks/CU
//            and strided, so each wave keeps several KiB in flight.
//  * generic — any strides (negative, permuted, up to 8-D) and any source dtype
//            via a wave-uniform dtype switch; div/mod index walk.
#include "md_hip.h"

// independent vectors per lane and trip in the fast kernels (A/B: profiles/r2_ew_unroll_ab.log)
#ifndef MD_EW_UNROLL
#define MD_EW_UNROLL 2      // launches that stay inside the Infinity Cache (128 MiB operands: 1, 2, 4 within noise)
#endif
#ifndef MD_EW_UNROLL_NT
#define MD_EW_UNROLL_NT 1   // non-temporal streams (> 320 MiB per launch): 400 MB multiply 214 us at 1, 238 at 2, 247 at 4
#endif

namespace {

// ------------------------------------------------------------------ generic ----
template <class F, class Tc, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_unary_generic(MdIter it, const void *x, int xdt, int x_scalar, Tc sx, To *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += stride) {
    int64_t offs[MD_MAX_OPS];
    md_iter_offsets(it, i, offs);
    Tc v = x_scalar ? sx : md_load<Tc>(x, xdt, offs[0]);
    out[offs[1]] = md_to_out<To>(F::apply(v));
  }
}

// ---- iteration spaces of three / four axes with a contiguous inner axis ---------------------------------------------------
// (B, R, C) * (B, 1, C), (B, R, C) * (1, R, 1), (N, C, H, W) * (1, C, 1, W), any call on a sliced 3-D view: the broadcasts of
// normalisation layers do not collapse to (rows, inner), so they took the element-by-element generic kernels (690-990 GB/s at
// 64 x 512 x 512). Here a lane owns one vector of four elements of the output's inner axis: the outer position from two or three
// divisions (32-bit when the launch allows), every operand read as a vector (inner stride 1) or one value (inner stride 0) in
// its own storage type, the result stored as a vector. Operands are of the loop's storage type (the mask of `where`: bool).
struct AxesGeom {
  int64_t e0, e1, e2, nv, rows;   // extents of the outer axes (e0 = 1 for three axes), vectors per inner row, e0 * e1 * e2
  int64_t st[3][3];               // operand k, outer axis j: stride in elements
  int in[3];                      // inner stride 0 / 1
};
template <class T, class C>
__device__ __forceinline__ void md_axes_load(const T *__restrict__ p, C s, int in, int64_t off, int64_t c, C (&r)[4]) {
  if (p == nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = s;
  } else if (in) {
    const MdVec<T, 4> t = *reinterpret_cast<const MdVec<T, 4> *>(p + off + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = md_cast<C>(t.v[j]);
  } else {
    const C t = md_cast<C>(p[off]);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = t;
  }
}
template <class F> struct AxUnary {
  template <class C0, class C1, class C2> static __device__ __forceinline__ auto apply(C0 a, C1, C2) { return F::apply(a); }
};
template <class F> struct AxBinary {
  template <class C0, class C1, class C2> static __device__ __forceinline__ auto apply(C0 a, C1 b, C2) { return F::apply(a, b); }
};
struct AxWhere {
  template <class C0, class C1, class C2> static __device__ __forceinline__ C1 apply(C0 c, C1 a, C2 b) { return c ? a : b; }
};
template <class Op, int NIN, class To, class T0, class C0, class T1, class C1, class T2, class C2>
__global__ void __launch_bounds__(MD_BLOCK) k_ew_axes(AxesGeom g, const T0 *__restrict__ p0, C0 s0, const T1 *__restrict__ p1, C1 s1,
                                                     const T2 *__restrict__ p2, C2 s2, To *__restrict__ out) {
  const int64_t total = g.rows * g.nv, gs = (int64_t)gridDim.x * blockDim.x;
  const bool narrow = total < (1ll << 31);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gs) {
    int64_t row, r0 = 0, r1, r2;
    if (narrow) {
      const uint32_t w = (uint32_t)v / (uint32_t)g.nv, q = w / (uint32_t)g.e2;
      row = w;
      r2 = w - q * (uint32_t)g.e2;
      r1 = q;
      if (g.e0 != 1) { const uint32_t t = q / (uint32_t)g.e1; r0 = t; r1 = q - t * (uint32_t)g.e1; }
    } else {
      row = v / g.nv;
      const int64_t q = row / g.e2;
      r2 = row - q * g.e2;
      r1 = q;
      if (g.e0 != 1) { r0 = q / g.e1; r1 = q - r0 * g.e1; }
    }
    const int64_t c = (v - row * g.nv) << 2;
    C0 x[4];
    C1 y[4] = {};
    C2 z[4] = {};
    md_axes_load(p0, s0, g.in[0], r0 * g.st[0][0] + r1 * g.st[0][1] + r2 * g.st[0][2], c, x);
    if constexpr (NIN > 1) md_axes_load(p1, s1, g.in[1], r0 * g.st[1][0] + r1 * g.st[1][1] + r2 * g.st[1][2], c, y);
    if constexpr (NIN > 2) md_axes_load(p2, s2, g.in[2], r0 * g.st[2][0] + r1 * g.st[2][1] + r2 * g.st[2][2], c, z);
    MdVec<To, 4> r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r.v[j] = md_to_out<To>(Op::apply(x[j], y[j], z[j]));
    *reinterpret_cast<MdVec<To, 4> *>(out + row * (g.nv << 2) + c) = r;
  }
}

template <class F, class Tc, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_binary_generic(MdIter it, const void *a, int adt, int a_scalar, Tc sa,
                                                            const void *b, int bdt, int b_scalar, Tc sb, To *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += stride) {
    int64_t offs[MD_MAX_OPS];
    md_iter_offsets(it, i, offs);
    Tc va = a_scalar ? sa : md_load<Tc>(a, adt, offs[0]);
    Tc vb = b_scalar ? sb : md_load<Tc>(b, bdt, offs[1]);
    out[offs[2]] = md_to_out<To>(F::apply(va, vb));
  }
}

template <class T>
__global__ void __launch_bounds__(MD_BLOCK) k_where_generic(MdIter it, const void *c, int cdt, int c_scalar, uint8_t sc,
                                                           const void *a, int adt, int a_scalar, T sa,
                                                           const void *b, int bdt, int b_scalar, T sb, T *out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += stride) {
    int64_t offs[MD_MAX_OPS];
    md_iter_offsets(it, i, offs);
    uint8_t vc = c_scalar ? sc : md_load<uint8_t>(c, cdt, offs[0]);
    T va = a_scalar ? sa : md_load<T>(a, adt, offs[1]);
    T vb = b_scalar ? sb : md_load<T>(b, bdt, offs[2]);
    out[offs[3]] = vc ? va : vb;
  }
}

// --------------------------------------------------------------------- fast ----
template <class T> struct FastOp {
  const T *p;   // nullptr -> host scalar
  int64_t os;   // outer (row) stride, elements
  int32_t is;   // inner stride: 0 or 1
};

template <class Tc> struct md_storage { using type = Tc; };
template <> struct md_storage<uint8_t> { using type = b8; };

// Streams larger than the 256 MiB Infinity Cache are touched once: a non-temporal hint on
// the 16-B loads and stores keeps them from allocating in L2/MALL (+10-12 % on 400 MB
// operands: profiles/r1_streaming_nt_ab.log). Smaller operands keep normal caching so that
// the consumer of a just-written intermediate (cfg4's z, masks) still hits the cache.
template <bool NT, class V> __device__ __forceinline__ V md_ld_stream(const V *p) {
  if constexpr (sizeof(V) == 16 && NT) {
    {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      i32x4 t = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(p));
      V v;
      __builtin_memcpy(&v, &t, 16);
      return v;
    }
  }
  return *p;
}
template <bool NT, class V> __device__ __forceinline__ void md_st_stream(V *p, const V &v) {
  if constexpr (sizeof(V) == 16 && NT) {
    {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      i32x4 t;
      __builtin_memcpy(&t, &v, 16);
      __builtin_nontemporal_store(t, reinterpret_cast<i32x4 *>(p));
      return;
    }
  }
  *p = v;
}

template <bool NT = false, class T, class Tc>
__device__ __forceinline__ void md_fast_load(const FastOp<T> &o, Tc s, int64_t row, int64_t c, Tc (&r)[4]) {
  if (o.p == nullptr) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = s;
  } else if (o.is) {
    MdVec<T, 4> v = md_ld_stream<NT>(reinterpret_cast<const MdVec<T, 4> *>(o.p + row * o.os + c));
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = md_cast<Tc>(v.v[j]);
  } else {
    Tc s1 = md_cast<Tc>(o.p[row * o.os]);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = s1;
  }
}
template <class T, class Tc>
__device__ __forceinline__ Tc md_fast_load1(const FastOp<T> &o, Tc s, int64_t row, int64_t c) {
  if (o.p == nullptr) return s;
  return md_cast<Tc>(o.p[row * o.os + (o.is ? c : 0)]);
}

// How an operand is read inside the loop is a COMPILE-TIME property of the kernel, so that the loads of one trip
// are unconditional and issue back to back (with the three-way run-time branch of md_fast_load around every
// load the compiler waited for each load before the next one: one load in flight per wave).
//   OM_VEC   unit inner stride: one 16-B (f32/i32) / 2 x 16-B (f64/i64) / 4-B (bool) load per vector
//   OM_SCAL  one value for the whole launch: a host scalar, or ONE device element behind a stride-0 view
//            (the 0-d seed of the backward pass broadcast to the gradient's shape) read once in the prologue
//   OM_FLEX  decided at run time per vector (column-broadcast operands: inner stride 0, row stride != 0)
enum { OM_FLEX = 0, OM_VEC = 1, OM_SCAL = 2 };

template <int MODE, bool NT, class T, class Tc>
__device__ __forceinline__ void md_op_load(const FastOp<T> &o, Tc s, int64_t row, int64_t c, Tc (&r)[4]) {
  if constexpr (MODE == OM_SCAL) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = s;
  } else if constexpr (MODE == OM_VEC) {
    MdVec<T, 4> v = md_ld_stream<NT>(reinterpret_cast<const MdVec<T, 4> *>(o.p + row * o.os + c));
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = md_cast<Tc>(v.v[j]);
  } else {
    md_fast_load<NT>(o, s, row, c, r);
  }
}
template <int MODE, class T, class Tc> __device__ __forceinline__ void md_op_prologue(const FastOp<T> &o, Tc &s) {
  if constexpr (MODE == OM_SCAL) {
    if (o.p != nullptr) s = md_cast<Tc>(o.p[0]);
  }
}

// Launch geometry of the fast kernels: vectors of 4 elements over (rows, inner); thread t takes vectors
// t, t + stride, t + 2 stride, ... . (dq, dr) = divmod(stride, inner / 4) from the host: the (row, column)
// position advances by additions, one division per thread instead of one per vector.
struct FastGrid {
  int64_t rows, inner, dq, dr;
};

// U independent vectors per lane and trip: all loads of the U vectors are issued before the first use.
template <int U, class Body> __device__ __forceinline__ void md_ew_drive(const Body &body, const FastGrid g) {
  const int64_t nv = g.inner >> 2, total = g.rows * nv;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t row = 0, cv = gid;
  if (g.rows != 1) {
    if ((uint64_t)gid < 0x100000000ull && (uint64_t)nv < 0x100000000ull) {
      row = (uint32_t)gid / (uint32_t)nv;
      cv = (uint32_t)gid - (uint32_t)row * (uint32_t)nv;
    } else {
      row = gid / nv;
      cv = gid - row * nv;
    }
  }
  auto advance = [&]() {
    cv += g.dr;
    row += g.dq;
    if (cv >= nv) { cv -= nv; ++row; }
  };
  int64_t v = gid;
  if constexpr (U > 1) {
    for (; v + (U - 1) * stride < total; v += U * stride) {
      typename Body::Regs r[U];
      int64_t off[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        body.load(r[u], row, cv << 2);
        off[u] = row * g.inner + (cv << 2);
        advance();
      }
#pragma unroll
      for (int u = 0; u < U; ++u) body.store(r[u], off[u]);
    }
  }
  for (; v < total; v += stride) {
    typename Body::Regs r;
    body.load(r, row, cv << 2);
    body.store(r, row * g.inner + (cv << 2));
    advance();
  }
  if (g.rows == 1) {  // 1-D: the up-to-3 elements behind the last whole vector
    const int64_t i = (nv << 2) + gid;
    if (i < g.inner) body.tail(i);
  }
}

// transcendental functors keep the ALU busy between memory operations and want all 8 blocks per CU; the cheap ones are
// pure streams and ran faster with 4 (400 MB copy 141 -> 132-134 us, square 138 -> 131; sin 137 -> 142 the other way)
template <class F> struct ew_heavy { static constexpr bool value = false; };
template <> struct ew_heavy<USin> { static constexpr bool value = true; };
template <> struct ew_heavy<UCos> { static constexpr bool value = true; };
template <> struct ew_heavy<UTan> { static constexpr bool value = true; };
template <> struct ew_heavy<USinh> { static constexpr bool value = true; };
template <> struct ew_heavy<UCosh> { static constexpr bool value = true; };
template <> struct ew_heavy<UTanh> { static constexpr bool value = true; };
template <> struct ew_heavy<UExp> { static constexpr bool value = true; };
template <> struct ew_heavy<ULog> { static constexpr bool value = true; };

template <class F, class Tc, class To, class Tx, int MX, bool NT> struct UnaryBody {
  static constexpr int kBlocksPerCU = ew_heavy<F>::value ? 8 : 4;
  FastOp<Tx> x;
  Tc sx;
  To *out;
  struct Regs { Tc x[4]; };
  __device__ __forceinline__ void prologue() { md_op_prologue<MX>(x, sx); }
  __device__ __forceinline__ void load(Regs &r, int64_t row, int64_t c) const { md_op_load<MX, NT>(x, sx, row, c, r.x); }
  __device__ __forceinline__ void store(const Regs &r, int64_t off) const {
    MdVec<To, 4> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = md_to_out<To>(F::apply(r.x[j]));
    md_st_stream<NT>(reinterpret_cast<MdVec<To, 4> *>(out + off), o);
  }
  __device__ __forceinline__ void tail(int64_t i) const { out[i] = md_to_out<To>(F::apply(md_fast_load1(x, sx, 0, i))); }
};

template <class F, class Tc, class To, class Ta, class Tb, int MA, int MB, bool NT> struct BinaryBody {
  // fewer resident waves ran faster for the binary forms — 400 MB multiply(x, y) 203 us at 8 blocks per CU, 195 at 4,
  // 190 at 2; cfg4 mask product (one streamed operand, f32 result) 31-35 -> 29.4-30.0, z > 0 28 -> 26 — while the unary
  // kernels lose below 8 (sin 137 -> 142 us): profiles/r2_ew_grid_ab.log
  static constexpr int kBlocksPerCU = 4;
  FastOp<Ta> a;
  FastOp<Tb> b;
  Tc sa, sb;
  To *out;
  struct Regs { Tc x[4], y[4]; };
  __device__ __forceinline__ void prologue() { md_op_prologue<MA>(a, sa); md_op_prologue<MB>(b, sb); }
  __device__ __forceinline__ void load(Regs &r, int64_t row, int64_t c) const {
    md_op_load<MA, NT>(a, sa, row, c, r.x);
    md_op_load<MB, NT>(b, sb, row, c, r.y);
  }
  __device__ __forceinline__ void store(const Regs &r, int64_t off) const {
    MdVec<To, 4> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = md_to_out<To>(F::apply(r.x[j], r.y[j]));
    md_st_stream<NT>(reinterpret_cast<MdVec<To, 4> *>(out + off), o);
  }
  __device__ __forceinline__ void tail(int64_t i) const {
    out[i] = md_to_out<To>(F::apply(md_fast_load1(a, sa, 0, i), md_fast_load1(b, sb, 0, i)));
  }
};

template <class T, class Tcnd, int MC, int MA, int MB> struct WhereBody {
  static constexpr int kBlocksPerCU = 8;
  FastOp<Tcnd> c;
  FastOp<T> a, b;
  uint8_t sc;
  T sa, sb;
  T *out;
  struct Regs { uint8_t c[4]; T x[4], y[4]; };
  __device__ __forceinline__ void prologue() { md_op_prologue<MC>(c, sc); md_op_prologue<MA>(a, sa); md_op_prologue<MB>(b, sb); }
  __device__ __forceinline__ void load(Regs &r, int64_t row, int64_t col) const {
    md_op_load<MC, false>(c, sc, row, col, r.c);
    md_op_load<MA, false>(a, sa, row, col, r.x);
    md_op_load<MB, false>(b, sb, row, col, r.y);
  }
  __device__ __forceinline__ void store(const Regs &r, int64_t off) const {
    MdVec<T, 4> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = r.c[j] ? r.x[j] : r.y[j];
    *reinterpret_cast<MdVec<T, 4> *>(out + off) = o;
  }
  __device__ __forceinline__ void tail(int64_t i) const {
    out[i] = md_fast_load1(c, sc, 0, i) ? md_fast_load1(a, sa, 0, i) : md_fast_load1(b, sb, 0, i);
  }
};

// One kernel for all three arities; the profiler shows the body type (operator, types, operand modes).
template <class Body, int U> __global__ void __launch_bounds__(MD_BLOCK) k_ew_fast(Body body, FastGrid g) {
  body.prologue();
  md_ew_drive<U>(body, g);
}

// ------------------------------------------------------------- transposed x ----
// out[b][r][c] = F(x[b*xb + c*xc + r]) : the operand is contiguous along the axis that is
// NOT out's contiguous one (x.T, swapaxes, transpose_grad: definitions.py:144-152,416-420).
// 64x64 tile through LDS: loads run along r (x's unit stride), stores along c (out's).
template <class F, class Tc, class To>
__global__ void __launch_bounds__(MD_BLOCK) k_unary_tr(const void *x, int xdt, int64_t xb, int64_t xc, To *out, int64_t R, int64_t Cn,
                                                      int tiles_r, int tiles_c) {
  __shared__ Tc tile[64][65];
  int64_t bid = blockIdx.x;
  const int64_t per = (int64_t)tiles_r * tiles_c;
  const int64_t b = bid / per;
  bid -= b * per;
  const int tr = (int)(bid / tiles_c), tc = (int)(bid - (int64_t)tr * tiles_c);
  const int64_t r0 = (int64_t)tr * 64, c0 = (int64_t)tc * 64;
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int c = w + 4 * k;
    if (r0 + l < R && c0 + c < Cn) tile[c][l] = md_load<Tc>(x, xdt, b * xb + (c0 + c) * xc + r0 + l);
  }
  __syncthreads();
  To *o = out + b * R * Cn;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int r = w + 4 * k;
    if (r0 + r < R && c0 + l < Cn) o[(r0 + r) * Cn + c0 + l] = md_to_out<To>(F::apply(tile[l][r]));
  }
}
// eligibility: out C-contiguous over (B, R, C) after the iterator's collapse, x unit-stride along R
static bool tr_geom(const MdIter &it, int xk, int ok, int64_t *B, int64_t *R, int64_t *Cn, int64_t *xb, int64_t *xc) {
  if (it.ndim != 2 && it.ndim != 3) return false;
  const int d0 = it.ndim - 2, d1 = it.ndim - 1;
  *R = it.shape[d0];
  *Cn = it.shape[d1];
  if (*R < 32 || *Cn < 32) return false;
  if (it.strides[ok][d1] != 1 || it.strides[ok][d0] != *Cn) return false;
  if (it.strides[xk][d0] != 1 || it.strides[xk][d1] < *R) return false;
  *xc = it.strides[xk][d1];
  *B = 1;
  *xb = 0;
  if (it.ndim == 3) {
    *B = it.shape[0];
    if (it.strides[ok][0] != *R * *Cn || it.strides[xk][0] < 0) return false;
    *xb = it.strides[xk][0];
  }
  const int64_t blocks = *B * ((*R + 63) / 64) * ((*Cn + 63) / 64);
  return blocks < (1ll << 31);
}

// ------------------------------------------------------- fast-path eligibility ----
struct FastGeom {
  int64_t rows, inner;
};
static bool fast_geom(const MdIter &it, int out_k, FastGeom *g) {
  if (it.ndim == 1) {
    if (it.strides[out_k][0] != 1) return false;
    g->rows = 1;
    g->inner = it.shape[0];
    return true;
  }
  if (it.ndim == 2) {
    if (it.strides[out_k][1] != 1 || it.strides[out_k][0] != it.shape[1]) return false;
    if (it.shape[1] & 3) return false;
    g->rows = it.shape[0];
    g->inner = it.shape[1];
    return true;
  }
  return false;
}
template <class T> static bool fast_aligned(const void *p) {
  const uintptr_t al = sizeof(T) * 4 > 16 ? 16 : sizeof(T) * 4;
  return ((uintptr_t)p % al) == 0;
}
template <class T> static bool fast_operand(const MdIter &it, int k, const mdhip_array *arr, const FastGeom &g, FastOp<T> *f) {
  if (arr->is_scalar) {
    f->p = nullptr; f->os = 0; f->is = 0;
    return true;
  }
  if (arr->dtype != md_dtype_of<T>::value) return false;
  const int64_t is = it.strides[k][it.ndim - 1];
  const int64_t os = it.ndim == 2 ? it.strides[k][0] : 0;
  if (is != 0 && is != 1) return false;
  if (is == 1) {
    if (!fast_aligned<T>(arr->data)) return false;
    if (g.rows > 1 && (os & 3)) return false;
  }
  f->p = (const T *)arr->data;
  f->os = os;
  f->is = (int32_t)is;
  return true;
}

// non-temporal streaming when one launch touches more than the Infinity Cache can hold
static int nt_for(int64_t bytes) {
  const int mode = (int)md_opt(MD_OPT_NT);  // 0 = never, 1 = always, -1 = by size
  if (mode >= 0) return mode != 0;
  return bytes > ((int64_t)320 << 20);
}

// compile-time read mode an operand qualifies for (OM_FLEX: only the run-time path can read it)
template <class T> static int fast_mode(const FastOp<T> &f, const FastGeom &g) {
  if (f.p == nullptr) return OM_SCAL;
  if (f.is) return OM_VEC;
  return (f.os == 0 || g.rows == 1) ? OM_SCAL : OM_FLEX;
}
template <class T> static int64_t fast_bytes(const FastOp<T> &f, const FastGeom &g) {
  if (f.p == nullptr || !f.is) return 0;
  return (f.os == 0 && g.rows > 1 ? g.inner : g.rows * g.inner) * (int64_t)sizeof(T);
}

template <class Body, int U> static int launch_fast(const Body &body, const FastGeom &g, const char *what) {
  const int64_t nv = g.inner >> 2;
  const int64_t work = g.rows * nv + (g.rows == 1 ? 4 : 0);
  const int cap = md_max_blocks();  // (MDHIP_MAX_BLOCKS overrides the per-kernel choice: experiments)
  const int grid = md_grid_for(work, MD_BLOCK, cap != MD_NUM_CUS * 8 ? cap : MD_NUM_CUS * Body::kBlocksPerCU);
  const int64_t stride = (int64_t)grid * MD_BLOCK;
  FastGrid fg;
  fg.rows = g.rows;
  fg.inner = g.inner;
  fg.dq = nv > 0 ? stride / nv : 0;
  fg.dr = nv > 0 ? stride % nv : 0;
  MD_LAUNCH((k_ew_fast<Body, U>), grid, MD_BLOCK, body, fg);
  return MD_LAUNCH_CHECK(what);
}

// k_ew_axes eligibility: three or four collapsed axes, a dense output, every array operand of the dtype the kernel reads it in with
// its inner axis contiguous (aligned for vector loads) or broadcast. `out_k` = the output's slot in the iterator.
static bool axes_geom(const MdIter &it, int out_k, const mdhip_array *out, size_t out_esize, const mdhip_array *const *ops, const int *dtypes,
                      const size_t *esizes, int n_in, AxesGeom *g) {
  if (it.ndim != 3 && it.ndim != 4) return false;
  const int nd = it.ndim, sh = 4 - nd;   // outer axis j of the kernel = iterator axis j - sh
  const int64_t inner = it.shape[nd - 1];
  if ((inner & 3) || it.total < (1 << 16) || it.strides[out_k][nd - 1] != 1) return false;
  int64_t dense = inner;
  for (int d = nd - 2; d >= 0; --d) {
    if (it.strides[out_k][d] != dense) return false;
    dense *= it.shape[d];
  }
  auto aligned = [](const void *p, size_t es) { return ((uintptr_t)p % (es * 4 > 16 ? 16 : es * 4)) == 0; };
  if (!aligned(out->data, out_esize)) return false;
  g->e0 = nd == 4 ? it.shape[0] : 1;
  g->e1 = it.shape[1 - sh];
  g->e2 = it.shape[2 - sh];
  g->nv = inner >> 2;
  g->rows = g->e0 * g->e1 * g->e2;
  for (int k = 0; k < 3; ++k) {
    g->in[k] = 0;
    for (int j = 0; j < 3; ++j) g->st[k][j] = 0;
  }
  bool any = false;
  for (int k = 0; k < n_in; ++k) {
    const mdhip_array *x = ops[k];
    if (x->is_scalar) continue;
    any = true;
    if (x->dtype != dtypes[k]) return false;
    const int64_t is = it.strides[k][nd - 1];
    if (is != 0 && is != 1) return false;
    g->in[k] = (int)is;
    for (int j = sh; j < 3; ++j) {
      const int64_t st = it.strides[k][j - sh];
      if (is == 1 && (st & 3)) return false;
      g->st[k][j] = st;
    }
    if (is == 1 && !aligned(x->data, esizes[k])) return false;
  }
  return any;
}

struct HipExec {
  // ------------------------------------------------------------------ unary ----
  template <class F, class Tc, class To, class Tx, int MX>
  static int unary_fast(const FastOp<Tx> &fx, Tc sx, To *out, const FastGeom &g, bool nt) {
    if (nt) return launch_fast<UnaryBody<F, Tc, To, Tx, MX, true>, MD_EW_UNROLL_NT>(UnaryBody<F, Tc, To, Tx, MX, true>{fx, sx, out}, g, "unary(fast)");
    return launch_fast<UnaryBody<F, Tc, To, Tx, MX, false>, MD_EW_UNROLL>(UnaryBody<F, Tc, To, Tx, MX, false>{fx, sx, out}, g, "unary(fast)");
  }
  template <class F, class Tc, class To>
  static int unary(const MdIter &it, const mdhip_array *x, const mdhip_array *out) {
    using Tx = typename md_storage<Tc>::type;
    Tc sx = x->is_scalar ? md_scalar_as<Tc>(x) : Tc();
    FastGeom g;
    FastOp<Tx> fx;
    if (fast_geom(it, 1, &g) && fast_aligned<To>(out->data) && fast_operand<Tx>(it, 0, x, g, &fx)) {
      const bool nt = nt_for(g.rows * g.inner * (int64_t)sizeof(To) + fast_bytes(fx, g));
      switch (fast_mode(fx, g)) {
        case OM_VEC: return unary_fast<F, Tc, To, Tx, OM_VEC>(fx, sx, (To *)out->data, g, nt);
        case OM_SCAL: return unary_fast<F, Tc, To, Tx, OM_SCAL>(fx, sx, (To *)out->data, g, nt);
        default: return launch_fast<UnaryBody<F, Tc, To, Tx, OM_FLEX, false>, 1>(UnaryBody<F, Tc, To, Tx, OM_FLEX, false>{fx, sx, (To *)out->data}, g, "unary(fast,flex)");
      }
    }
    // dtype conversions (astype, definitions.py:429-432; raw .astype, tensor.py:105): the same streaming
    // kernel with the SOURCE storage type as the load type
    if constexpr (md_same<F, UCopy>::value) {
      if (!x->is_scalar && x->dtype != md_dtype_of<Tx>::value && fast_geom(it, 1, &g) && fast_aligned<To>(out->data)) {
#define MD_CAST_FROM(S)                                                                                                   \
        {                                                                                                                 \
          FastOp<S> fs;                                                                                                   \
          if (fast_operand<S>(it, 0, x, g, &fs) && fast_mode(fs, g) == OM_VEC)                                            \
            return launch_fast<UnaryBody<F, Tc, To, S, OM_VEC, false>, MD_EW_UNROLL>(UnaryBody<F, Tc, To, S, OM_VEC, false>{fs, sx, (To *)out->data}, g, "unary(fast,cast)"); \
        }
        switch (x->dtype) {
          case MDHIP_F32: MD_CAST_FROM(float) break;
          case MDHIP_F64: MD_CAST_FROM(double) break;
          case MDHIP_I64: MD_CAST_FROM(int64_t) break;
          case MDHIP_I32: MD_CAST_FROM(int32_t) break;
          case MDHIP_BOOL: MD_CAST_FROM(b8) break;
        }
#undef MD_CAST_FROM
      }
    }
    {
      AxesGeom ag;
      const mdhip_array *ops[1] = {x};
      const int dts[1] = {md_dtype_of<Tx>::value};
      const size_t ess[1] = {sizeof(Tx)};
      if (axes_geom(it, 1, out, sizeof(To), ops, dts, ess, 1, &ag)) {
        MD_LAUNCH((k_ew_axes<AxUnary<F>, 1, To, Tx, Tc, Tx, Tc, Tx, Tc>), md_grid_for(it.total >> 2), MD_BLOCK, ag, (const Tx *)x->data, sx, (const Tx *)nullptr, sx,
                  (const Tx *)nullptr, sx, (To *)out->data);
        return MD_LAUNCH_CHECK("unary(axes, vectors)");
      }
    }
    int64_t B, R, Cn, xb, xc;
    if (!x->is_scalar && it.total >= (1 << 14) && tr_geom(it, 0, 1, &B, &R, &Cn, &xb, &xc)) {
      const int tr = (int)((R + 63) / 64), tc = (int)((Cn + 63) / 64);
      MD_LAUNCH((k_unary_tr<F, Tc, To>), (unsigned)(B * tr * tc), MD_BLOCK, x->data, x->dtype, xb, xc, (To *)out->data, R, Cn, tr, tc);
      return MD_LAUNCH_CHECK("unary(transposed)");
    }
    MD_LAUNCH((k_unary_generic<F, Tc, To>), md_grid_for(it.total), MD_BLOCK, it, x->data, x->dtype, x->is_scalar, sx, (To *)out->data);
    return MD_LAUNCH_CHECK("unary(generic)");
  }

  // ----------------------------------------------------------------- binary ----
  template <class F, class Tc, class To, class Ta, class Tb, int MA, int MB>
  static int binary_fast(const FastOp<Ta> &fa, const FastOp<Tb> &fb, Tc sa, Tc sb, To *out, const FastGeom &g, bool nt) {
    if (nt) return launch_fast<BinaryBody<F, Tc, To, Ta, Tb, MA, MB, true>, MD_EW_UNROLL_NT>(BinaryBody<F, Tc, To, Ta, Tb, MA, MB, true>{fa, fb, sa, sb, out}, g, "binary(fast)");
    return launch_fast<BinaryBody<F, Tc, To, Ta, Tb, MA, MB, false>, MD_EW_UNROLL>(BinaryBody<F, Tc, To, Ta, Tb, MA, MB, false>{fa, fb, sa, sb, out}, g, "binary(fast)");
  }
  template <class F, class Tc, class To, class Ta, class Tb>
  static bool try_binary_fast(const MdIter &it, const FastGeom &g, const mdhip_array *a, const mdhip_array *b,
                              const mdhip_array *out, Tc sa, Tc sb, int *status) {
    FastOp<Ta> fa;
    FastOp<Tb> fb;
    if (!fast_operand<Ta>(it, 0, a, g, &fa) || !fast_operand<Tb>(it, 1, b, g, &fb)) return false;
    const bool nt = nt_for(g.rows * g.inner * (int64_t)sizeof(To) + fast_bytes(fa, g) + fast_bytes(fb, g));
    To *o = (To *)out->data;
    const int ma = fast_mode(fa, g), mb = fast_mode(fb, g);
    // the three streaming forms get their own kernels for the float loops (everything BASELINE's graphs launch);
    // integer loops and column-broadcast operands take the run-time path
    if constexpr (md_is_float<Tc>::value) {
      if (ma == OM_VEC && mb == OM_VEC) { *status = binary_fast<F, Tc, To, Ta, Tb, OM_VEC, OM_VEC>(fa, fb, sa, sb, o, g, nt); return true; }
      if (ma == OM_VEC && mb == OM_SCAL) { *status = binary_fast<F, Tc, To, Ta, Tb, OM_VEC, OM_SCAL>(fa, fb, sa, sb, o, g, nt); return true; }
      if (ma == OM_SCAL && mb == OM_VEC) { *status = binary_fast<F, Tc, To, Ta, Tb, OM_SCAL, OM_VEC>(fa, fb, sa, sb, o, g, nt); return true; }
    }
    using Body = BinaryBody<F, Tc, To, Ta, Tb, OM_FLEX, OM_FLEX, false>;
    *status = launch_fast<Body, 1>(Body{fa, fb, sa, sb, o}, g, "binary(fast,flex)");
    return true;
  }
  template <class F, class Tc, class To>
  static int binary(const MdIter &it, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
    using Ts = typename md_storage<Tc>::type;
    Tc sa = a->is_scalar ? md_scalar_as<Tc>(a) : Tc();
    Tc sb = b->is_scalar ? md_scalar_as<Tc>(b) : Tc();
    FastGeom g;
    if (!(a->is_scalar && b->is_scalar) && fast_geom(it, 2, &g) && fast_aligned<To>(out->data)) {
      int status = MDHIP_OK;
      if (try_binary_fast<F, Tc, To, Ts, Ts>(it, g, a, b, out, sa, sb, &status)) return status;
      // bool mask times float payload (relu / where gradients: definitions.py:555-559)
      if constexpr (md_same<F, BMul>::value && md_is_float<Tc>::value) {
        if (try_binary_fast<F, Tc, To, Ts, b8>(it, g, a, b, out, sa, sb, &status)) return status;
        if (try_binary_fast<F, Tc, To, b8, Ts>(it, g, a, b, out, sa, sb, &status)) return status;
      }
    }
    if (!(a->is_scalar && b->is_scalar)) {
      // three / four collapsed axes, the output dense, the inner axis contiguous (or broadcast) in every operand: vectors of four
      AxesGeom ag;
      const mdhip_array *ops[2] = {a, b};
      const int dts[2] = {md_dtype_of<Ts>::value, md_dtype_of<Ts>::value};
      const size_t ess[2] = {sizeof(Ts), sizeof(Ts)};
      if (axes_geom(it, 2, out, sizeof(To), ops, dts, ess, 2, &ag)) {
        MD_LAUNCH((k_ew_axes<AxBinary<F>, 2, To, Ts, Tc, Ts, Tc, Ts, Tc>), md_grid_for(it.total >> 2), MD_BLOCK, ag, a->is_scalar ? nullptr : (const Ts *)a->data, sa,
                  b->is_scalar ? nullptr : (const Ts *)b->data, sb, (const Ts *)nullptr, sb, (To *)out->data);
        return MD_LAUNCH_CHECK("binary(axes, vectors)");
      }
    }
    // (MD_LAUNCH everywhere a call's main kernel starts: bench.py's attached events then time THAT kernel; a path that ignored them
    // sent the whole tag to marker brackets, ~5 us long on a 30-us kernel)
    MD_LAUNCH((k_binary_generic<F, Tc, To>), md_grid_for(it.total), MD_BLOCK, it, a->data, a->dtype, a->is_scalar, sa, b->data, b->dtype, b->is_scalar, sb,
              (To *)out->data);
    return MD_LAUNCH_CHECK("binary(generic)");
  }

  // ------------------------------------------------------------------ where ----
  template <class T>
  static int where(const MdIter &it, const mdhip_array *c, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
    uint8_t sc = c->is_scalar ? md_scalar_as<uint8_t>(c) : 0;
    T sa = a->is_scalar ? md_scalar_as<T>(a) : T();
    T sb = b->is_scalar ? md_scalar_as<T>(b) : T();
    FastGeom g;
    FastOp<b8> fc;
    FastOp<T> fa, fb;
    if (fast_geom(it, 3, &g) && fast_aligned<T>(out->data) && fast_operand<b8>(it, 0, c, g, &fc) &&
        fast_operand<T>(it, 1, a, g, &fa) && fast_operand<T>(it, 2, b, g, &fb)) {
      T *o = (T *)out->data;
      const int mc = fast_mode(fc, g), ma = fast_mode(fa, g), mb = fast_mode(fb, g);
#define MD_WHERE_FORM(MA_, MB_)                                                                                             \
      if (mc == OM_VEC && ma == MA_ && mb == MB_) {                                                                         \
        using Body = WhereBody<T, b8, OM_VEC, MA_, MB_>;                                                                    \
        return launch_fast<Body, MD_EW_UNROLL>(Body{fc, fa, fb, sc, sa, sb, o}, g, "where(fast)");                          \
      }
      MD_WHERE_FORM(OM_VEC, OM_SCAL)   // relu := where(z > 0, z, 0)
      MD_WHERE_FORM(OM_SCAL, OM_VEC)
      MD_WHERE_FORM(OM_VEC, OM_VEC)
      MD_WHERE_FORM(OM_SCAL, OM_SCAL)  // mod_grad: where(x % y == 0, 0, grad) with a broadcast seed
#undef MD_WHERE_FORM
      using Body = WhereBody<T, b8, OM_FLEX, OM_FLEX, OM_FLEX>;
      return launch_fast<Body, 1>(Body{fc, fa, fb, sc, sa, sb, o}, g, "where(fast,flex)");
    }
    {
      AxesGeom ag;
      const mdhip_array *ops[3] = {c, a, b};
      const int dts[3] = {MDHIP_BOOL, md_dtype_of<T>::value, md_dtype_of<T>::value};
      const size_t ess[3] = {1, sizeof(T), sizeof(T)};
      if (axes_geom(it, 3, out, sizeof(T), ops, dts, ess, 3, &ag)) {
        MD_LAUNCH((k_ew_axes<AxWhere, 3, T, b8, uint8_t, T, T, T, T>), md_grid_for(it.total >> 2), MD_BLOCK, ag, c->is_scalar ? nullptr : (const b8 *)c->data, sc,
                  a->is_scalar ? nullptr : (const T *)a->data, sa, b->is_scalar ? nullptr : (const T *)b->data, sb, (T *)out->data);
        return MD_LAUNCH_CHECK("where(axes, vectors)");
      }
    }
    MD_LAUNCH((k_where_generic<T>), md_grid_for(it.total), MD_BLOCK, it, c->data, c->dtype, c->is_scalar, sc, a->data, a->dtype, a->is_scalar, sa, b->data,
              b->dtype, b->is_scalar, sb, (T *)out->data);
    return MD_LAUNCH_CHECK("where(generic)");
  }
};

template <class T> __global__ void __launch_bounds__(MD_BLOCK) k_arange(T *out, int64_t n, int64_t stride, double start, double step) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
    if constexpr (md_is_float<T>::value) out[i * stride] = (T)(start + (double)i * step);
    else out[i * stride] = (T)((int64_t)start + i * (int64_t)step);
  }
}

// ------------------------------------------------------------- any -> any conversion ----
// The kernel behind the storage-only dtypes (include/mdhip.h: int8/16, uint8/16/32/64, float16): loads through the source's
// carrier (int64 / uint64 / double), stores with the destination's conversion; the dtype switches are wave-uniform. Off the
// BASELINE paths (nothing there uses a narrow type): one generic strided walk.
template <class C>
__global__ void __launch_bounds__(MD_BLOCK) k_convert(MdIter it, const void *x, int xdt, void *out, int odt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += stride) {
    int64_t offs[MD_MAX_OPS];
    md_iter_offsets(it, i, offs);
    md_store_any<C>(out, odt, offs[1], md_load_any<C>(x, xdt, offs[0]));
  }
}

}  // namespace

// narrow.hip: the same entry points when a storage-only dtype (int8/16, uint8/16/32/64, float16) takes part
int md_narrow_unary(int op, const mdhip_array *x, const mdhip_array *out);
int md_narrow_binary(int op, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out, int cdt);
int md_narrow_where(const mdhip_array *c, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out);
static inline bool md_narrow_code(int dt) { return dt >= MDHIP_NUM_DTYPES && dt < MDHIP_NUM_ALL_DTYPES; }
static inline bool md_narrow_arr(const mdhip_array *a) { return a && !a->is_scalar && md_narrow_code(a->dtype); }

extern "C" {

int mdhip_unary(int op, const mdhip_array *x, const mdhip_array *out) {
  if (op != MDHIP_U_COPY && x && out && (md_narrow_arr(x) || md_narrow_arr(out))) return md_narrow_unary(op, x, out);
  return md_unary_dispatch<HipExec>(op, x, out);
}

int mdhip_convert(const mdhip_array *x, const mdhip_array *out) {
  MD_TRY(md_check_any_array(x, "convert x"));
  MD_TRY(md_check_any_array(out, "convert out"));
  MdIter it;
  const mdhip_array *ops[2] = {x, out};
  MD_TRY(md_build_iter(&it, 2, ops, out));
  if (it.total == 0) return MDHIP_OK;
  const int grid = md_grid_for(it.total);
  switch (md_dtype_carrier(x->dtype)) {
    case 2: k_convert<double><<<grid, MD_BLOCK, 0, md_stream()>>>(it, x->data, x->dtype, out->data, out->dtype); break;
    case 1: k_convert<uint64_t><<<grid, MD_BLOCK, 0, md_stream()>>>(it, x->data, x->dtype, out->data, out->dtype); break;
    default: k_convert<int64_t><<<grid, MD_BLOCK, 0, md_stream()>>>(it, x->data, x->dtype, out->data, out->dtype); break;
  }
  return MD_LAUNCH_CHECK("convert");
}

int mdhip_binary(int op, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out, int cdt) {
  if (a && b && out && (md_narrow_arr(a) || md_narrow_arr(b) || md_narrow_arr(out) || md_narrow_code(cdt))) return md_narrow_binary(op, a, b, out, cdt);
  return md_binary_dispatch<HipExec>(op, a, b, out, cdt);
}

int mdhip_where(const mdhip_array *c, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
  if (c && a && b && out && (md_narrow_arr(c) || md_narrow_arr(a) || md_narrow_arr(b) || md_narrow_arr(out))) return md_narrow_where(c, a, b, out);
  return md_where_dispatch<HipExec>(c, a, b, out);
}

int mdhip_fill(const mdhip_array *out, const mdhip_array *scalar) {
  if (!scalar || !scalar->is_scalar) return md_fail(MDHIP_EVALUE, "fill: value must be a scalar descriptor");
  return md_unary_dispatch<HipExec>(MDHIP_U_COPY, scalar, out);
}

int mdhip_arange(const mdhip_array *out, double start, double step) {
  MD_TRY(md_check_array(out, "arange out"));
  if (out->ndim != 1) return md_fail(MDHIP_EVALUE, "arange: out must be 1-D");
  const int64_t n = out->shape[0];
  if (n == 0) return MDHIP_OK;
  const int grid = md_grid_for(n);
  switch (out->dtype) {
    case MDHIP_I32: k_arange<int32_t><<<grid, MD_BLOCK, 0, md_stream()>>>((int32_t *)out->data, n, out->strides[0], start, step); break;
    case MDHIP_I64: k_arange<int64_t><<<grid, MD_BLOCK, 0, md_stream()>>>((int64_t *)out->data, n, out->strides[0], start, step); break;
    case MDHIP_F32: k_arange<float><<<grid, MD_BLOCK, 0, md_stream()>>>((float *)out->data, n, out->strides[0], start, step); break;
    case MDHIP_F64: k_arange<double><<<grid, MD_BLOCK, 0, md_stream()>>>((double *)out->data, n, out->strides[0], start, step); break;
    default: return md_fail(MDHIP_ETYPE, "arange: unsupported dtype %s", md_dtype_name(out->dtype));
  }
  return MD_LAUNCH_CHECK("arange");
}

}  // extern "C"
