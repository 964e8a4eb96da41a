// narrow.hip — elementwise kernels for the storage-only dtypes (int8/16, uint8/16/32/64, float16) and for mixed calls that
// involve one: ONE launch, every operand read once in its own storage type, the result written once in its own
// (round 3 ran these as promote -> wide kernel -> demote: three launches and >= 3x the algorithmic traffic).
//
// Serves the same backend names as elementwise.hip (reference minidiff/backend/numpy.py:19-95) for the dtypes of
// numpy.py:188-200 beyond bool / int32 / int64 / float32 / float64. Semantics and the carrier types: md_narrow.h.
//
// Two kernel families:
//  * stream  — every array operand contiguous, of ONE storage type S (the loop dtype's), the other operand possibly a scalar;
//              output contiguous of type S (arithmetic) or bool (comparisons). 16 B per lane and load — 16 int8, 8 float16 /
//              int16, 4 uint32, 2 uint64 elements — two vectors per lane and trip, grid-stride, non-temporal above the
//              Infinity Cache. HBM-bound at (operand bytes + result bytes): int8 * int8 moves 3 bytes per element.
//  * generic — any strides, any mix of the twelve dtypes: wave-uniform dtype switches around a div/mod walk.
#include "md_hip.h"
#include "md_narrow.h"

namespace {

enum { NM_VEC = 1, NM_SCAL = 2 };

template <bool NT, class V> __device__ __forceinline__ V nw_ld(const V *p) {
  if constexpr (sizeof(V) == 16 && NT) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 t = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(p));
    V v;
    __builtin_memcpy(&v, &t, 16);
    return v;
  } else {
    return *p;
  }
}
template <bool NT, class V> __device__ __forceinline__ void nw_st(V *p, const V &v) {
  if constexpr (sizeof(V) == 16 && NT) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 t;
    __builtin_memcpy(&t, &v, 16);
    __builtin_nontemporal_store(t, reinterpret_cast<i32x4 *>(p));
  } else if constexpr (sizeof(V) == 8 && NT) {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    i32x2 t;
    __builtin_memcpy(&t, &v, 8);
    __builtin_nontemporal_store(t, reinterpret_cast<i32x2 *>(p));
  } else {
    *p = v;
  }
}

// result storage of functor F over carrier Tc for loop storage S: S itself, or bool for the comparison / logical family
template <class F, class Tc, class S> struct nw_out {
  using R = decltype(F::apply(Tc(), Tc()));
  using type = typename md_cond<md_same<R, b8>::value, b8, S>::type;
};
template <class F, class Tc, class S> struct nw_out1 {
  using R = decltype(F::apply(Tc()));
  using type = typename md_cond<md_same<R, b8>::value, b8, S>::type;
};

// ------------------------------------------------------- broadcasts, 16-B vectors ----
// Two to four collapsed axes with the inner axis contiguous or broadcast in every operand ((R, C) + (C,), (B, R, C) * (B, 1, C) in
// float16 / int8 ..): a lane owns one 16-B vector of the OUTPUT's storage type count (8 float16, 16 int8 ..); the scheme of
// elementwise.hip's k_ew_axes. Was the one-element generic kernel (~350 GB/s).
struct NwAxes {
  int64_t e0, e1, e2, nv, rows;
  int64_t st[2][3];
  int in[2];
};
template <class F, class Tc, class S>
__global__ void __launch_bounds__(MD_BLOCK) k_nw_binary_axes(NwAxes g, const S *__restrict__ a, const S *__restrict__ b, Tc sa, Tc sb,
                                                            typename nw_out<F, Tc, S>::type *__restrict__ out) {
  using So = typename nw_out<F, Tc, S>::type;
  constexpr int E = 16 / sizeof(S);
  typedef MdVec<S, E> Vin;
  typedef MdVec<So, E> Vout;
  const int64_t total = g.rows * g.nv, gs = (int64_t)gridDim.x * MD_BLOCK;
  auto load = [&](const S *p, Tc s, int in, int64_t off, int64_t c, Tc (&r)[E]) {
    if (p == nullptr) {
#pragma unroll
      for (int j = 0; j < E; ++j) r[j] = s;
    } else if (in) {
      const Vin t = *reinterpret_cast<const Vin *>(p + off + c);
#pragma unroll
      for (int j = 0; j < E; ++j) r[j] = md_cast<Tc>(t.v[j]);
    } else {
      const Tc t = md_cast<Tc>(p[off]);
#pragma unroll
      for (int j = 0; j < E; ++j) r[j] = t;
    }
  };
  for (int64_t v = (int64_t)blockIdx.x * MD_BLOCK + threadIdx.x; v < total; v += gs) {
    const int64_t row = v / g.nv, q = row / g.e2, r2 = row - q * g.e2;
    int64_t r0 = 0, r1 = q;
    if (g.e0 != 1) { r0 = q / g.e1; r1 = q - r0 * g.e1; }
    const int64_t c = (v - row * g.nv) * E;
    Tc x[E], y[E];
    load(a, sa, g.in[0], r0 * g.st[0][0] + r1 * g.st[0][1] + r2 * g.st[0][2], c, x);
    load(b, sb, g.in[1], r0 * g.st[1][0] + r1 * g.st[1][1] + r2 * g.st[1][2], c, y);
    Vout o;
#pragma unroll
    for (int j = 0; j < E; ++j) o.v[j] = md_cast<So>(F::apply(x[j], y[j]));
    *reinterpret_cast<Vout *>(out + row * (g.nv * E) + c) = o;
  }
}

// ------------------------------------------------------------------- stream ----
template <class F, class Tc, class S, int MA, int MB, bool NT>
__global__ void __launch_bounds__(MD_BLOCK) k_nw_binary(const S *__restrict__ a, const S *__restrict__ b, Tc sa, Tc sb,
                                                       typename nw_out<F, Tc, S>::type *__restrict__ out, int64_t n) {
  using So = typename nw_out<F, Tc, S>::type;
  constexpr int E = 16 / sizeof(S);
  typedef MdVec<S, E> Vin;
  typedef MdVec<So, E> Vout;
  const int64_t nv = n / E, gs = (int64_t)gridDim.x * MD_BLOCK, gid = (int64_t)blockIdx.x * MD_BLOCK + threadIdx.x;
  const Vin *pa = reinterpret_cast<const Vin *>(a), *pb = reinterpret_cast<const Vin *>(b);
  Vout *po = reinterpret_cast<Vout *>(out);
  auto one = [&](const Vin &va, const Vin &vb, int64_t i) {
    Vout o;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const Tc x = MA == NM_VEC ? md_cast<Tc>(va.v[j]) : sa, y = MB == NM_VEC ? md_cast<Tc>(vb.v[j]) : sb;
      o.v[j] = md_cast<So>(F::apply(x, y));
    }
    nw_st<NT>(po + i, o);
  };
  int64_t i = gid;
  // two vectors per operand in flight on the cached path, ONE on the non-temporal path (as the wide kernels, elementwise.hip
  // MD_EW_UNROLL_NT: more in flight ran slower on streams that bypass the caches — here 59.5 -> 6x % for int8 * int8 on 2**30 elements)
  if constexpr (!NT) {
    for (; i + gs < nv; i += 2 * gs) {
      Vin a0, a1, b0, b1;
      if constexpr (MA == NM_VEC) { a0 = nw_ld<NT>(pa + i); a1 = nw_ld<NT>(pa + i + gs); }
      if constexpr (MB == NM_VEC) { b0 = nw_ld<NT>(pb + i); b1 = nw_ld<NT>(pb + i + gs); }
      one(a0, b0, i);
      one(a1, b1, i + gs);
    }
  }
  for (; i < nv; i += gs) {
    Vin a0, b0;
    if constexpr (MA == NM_VEC) a0 = nw_ld<NT>(pa + i);
    if constexpr (MB == NM_VEC) b0 = nw_ld<NT>(pb + i);
    one(a0, b0, i);
  }
  const int64_t t = nv * E + gid;   // the up-to-(E-1) elements behind the last whole vector
  if (t < n) {
    const Tc x = MA == NM_VEC ? md_cast<Tc>(a[t]) : sa, y = MB == NM_VEC ? md_cast<Tc>(b[t]) : sb;
    out[t] = md_cast<So>(F::apply(x, y));
  }
}

template <class F, class Tc, class S, bool NT>
__global__ void __launch_bounds__(MD_BLOCK) k_nw_unary(const S *__restrict__ x, typename nw_out1<F, Tc, S>::type *__restrict__ out, int64_t n) {
  using So = typename nw_out1<F, Tc, S>::type;
  constexpr int E = 16 / sizeof(S);
  typedef MdVec<S, E> Vin;
  typedef MdVec<So, E> Vout;
  const int64_t nv = n / E, gs = (int64_t)gridDim.x * MD_BLOCK, gid = (int64_t)blockIdx.x * MD_BLOCK + threadIdx.x;
  const Vin *px = reinterpret_cast<const Vin *>(x);
  Vout *po = reinterpret_cast<Vout *>(out);
  auto one = [&](const Vin &v, int64_t i) {
    Vout o;
#pragma unroll
    for (int j = 0; j < E; ++j) o.v[j] = md_cast<So>(F::apply(md_cast<Tc>(v.v[j])));
    nw_st<NT>(po + i, o);
  };
  int64_t i = gid;
  if constexpr (!NT) {
    for (; i + gs < nv; i += 2 * gs) {
      const Vin v0 = nw_ld<NT>(px + i), v1 = nw_ld<NT>(px + i + gs);
      one(v0, i);
      one(v1, i + gs);
    }
  }
  for (; i < nv; i += gs) one(nw_ld<NT>(px + i), i);
  const int64_t t = nv * E + gid;
  if (t < n) out[t] = md_cast<So>(F::apply(md_cast<Tc>(x[t])));
}

// ------------------------------------------------------------------ generic ----
template <class F, class Tc>
__global__ void __launch_bounds__(MD_BLOCK) k_nw_unary_generic(MdIter it, const void *x, int xdt, int x_scalar, Tc sx, void *out, int odt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += stride) {
    int64_t offs[MD_MAX_OPS];
    md_iter_offsets(it, i, offs);
    const Tc v = x_scalar ? sx : md_load<Tc>(x, xdt, offs[0]);
    md_store_as(out, odt, offs[1], F::apply(v));
  }
}
template <class F, class Tc>
__global__ void __launch_bounds__(MD_BLOCK) k_nw_binary_generic(MdIter it, const void *a, int adt, int a_scalar, Tc sa, const void *b, int bdt,
                                                               int b_scalar, Tc sb, void *out, int odt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += stride) {
    int64_t offs[MD_MAX_OPS];
    md_iter_offsets(it, i, offs);
    const Tc va = a_scalar ? sa : md_load<Tc>(a, adt, offs[0]);
    const Tc vb = b_scalar ? sb : md_load<Tc>(b, bdt, offs[1]);
    md_store_as(out, odt, offs[2], F::apply(va, vb));
  }
}
template <class Tc>
__global__ void __launch_bounds__(MD_BLOCK) k_nw_where_generic(MdIter it, const void *c, int cdt, int c_scalar, uint8_t sc, const void *a, int adt,
                                                              int a_scalar, Tc sa, const void *b, int bdt, int b_scalar, Tc sb, void *out, int odt) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < it.total; i += stride) {
    int64_t offs[MD_MAX_OPS];
    md_iter_offsets(it, i, offs);
    const uint8_t vc = c_scalar ? sc : md_load<uint8_t>(c, cdt, offs[0]);
    const Tc va = a_scalar ? sa : md_load<Tc>(a, adt, offs[1]);
    const Tc vb = b_scalar ? sb : md_load<Tc>(b, bdt, offs[2]);
    md_store_as(out, odt, offs[3], vc ? va : vb);
  }
}

// ---------------------------------------------------------------- eligibility ----
// the whole call as ONE contiguous run: every array operand dense over the collapsed 1-D space, 16-B aligned
static bool nw_contiguous(const MdIter &it, int nops, const mdhip_array *const *ops) {
  if (it.ndim != 1) return false;
  for (int k = 0; k < nops; ++k) {
    if (ops[k]->is_scalar) continue;
    if (it.strides[k][0] != 1) return false;
    if ((uintptr_t)ops[k]->data & 15) return false;
  }
  return true;
}
static bool nw_nt(int64_t bytes) {
  const int mode = (int)md_opt(MD_OPT_NT);
  if (mode >= 0) return mode != 0;
  return bytes > ((int64_t)320 << 20);
}
static int nw_grid(int64_t vectors, int per_cu) { return md_grid_for(vectors + 1, MD_BLOCK, MD_NUM_CUS * per_cu); }

// storage type of a loop dtype, as a type
template <int DT> struct nw_storage;
template <> struct nw_storage<MDHIP_I8> { using type = int8_t; };
template <> struct nw_storage<MDHIP_I16> { using type = int16_t; };
template <> struct nw_storage<MDHIP_U8> { using type = uint8_t; };
template <> struct nw_storage<MDHIP_U16> { using type = uint16_t; };
template <> struct nw_storage<MDHIP_U32> { using type = uint32_t; };
template <> struct nw_storage<MDHIP_U64> { using type = uint64_t; };
template <> struct nw_storage<MDHIP_F16> { using type = f16; };

struct HipExecN {
  // -------------------------------------------------------------------- unary ----
  template <class F, class Tc, int DT> static int unary_stream(const mdhip_array *x, const mdhip_array *out, int64_t n) {
    using S = typename nw_storage<DT>::type;
    using So = typename nw_out1<F, Tc, S>::type;
    const bool nt = nw_nt(n * (int64_t)(sizeof(S) + sizeof(So)));
    const int grid = nw_grid(n / (16 / (int64_t)sizeof(S)), 8);
    if (nt) MD_LAUNCH((k_nw_unary<F, Tc, S, true>), grid, MD_BLOCK, (const S *)x->data, (So *)out->data, n);
    else MD_LAUNCH((k_nw_unary<F, Tc, S, false>), grid, MD_BLOCK, (const S *)x->data, (So *)out->data, n);
    return MD_LAUNCH_CHECK("unary(narrow, stream)");
  }
  template <class F, class Tc> static int nunary(const MdIter &it, const mdhip_array *x, const mdhip_array *out) {
    using R = decltype(F::apply(Tc()));
    const mdhip_array *ops[2] = {x, out};
    // stream form: x and out of the same storage-only type (or out bool), dense
    if (!x->is_scalar && md_is_narrow(x->dtype) && nw_contiguous(it, 2, ops) && (md_same<R, b8>::value ? out->dtype == MDHIP_BOOL : out->dtype == x->dtype)) {
      switch (x->dtype) {
#define MD_NW_U(DT)                                                                                                          \
  case DT:                                                                                                                   \
    if constexpr (md_same<Tc, typename md_carrier_type<DT>::type>::value) return unary_stream<F, Tc, DT>(x, out, it.total);  \
    break;
        MD_NW_U(MDHIP_I8) MD_NW_U(MDHIP_I16) MD_NW_U(MDHIP_U8) MD_NW_U(MDHIP_U16) MD_NW_U(MDHIP_U32) MD_NW_U(MDHIP_U64) MD_NW_U(MDHIP_F16)
#undef MD_NW_U
      }
    }
    const Tc sx = x->is_scalar ? md_scalar_as<Tc>(x) : Tc();
    k_nw_unary_generic<F, Tc><<<md_grid_for(it.total), MD_BLOCK, 0, md_stream()>>>(it, x->data, x->dtype, x->is_scalar, sx, out->data, out->dtype);
    return MD_LAUNCH_CHECK("unary(narrow, generic)");
  }

  // ------------------------------------------------------------------- binary ----
  template <class F, class Tc, int DT, int MA, int MB>
  static int binary_stream(const mdhip_array *a, const mdhip_array *b, Tc sa, Tc sb, const mdhip_array *out, int64_t n) {
    using S = typename nw_storage<DT>::type;
    using So = typename nw_out<F, Tc, S>::type;
    const int64_t bytes = n * (int64_t)(sizeof(So) + (MA == NM_VEC ? sizeof(S) : 0) + (MB == NM_VEC ? sizeof(S) : 0));
    const bool nt = nw_nt(bytes);
    const int grid = nw_grid(n / (16 / (int64_t)sizeof(S)), MA == NM_VEC && MB == NM_VEC ? 4 : 8);   // (as the wide kernels: three streams want fewer resident waves)
    if (nt) MD_LAUNCH((k_nw_binary<F, Tc, S, MA, MB, true>), grid, MD_BLOCK, (const S *)a->data, (const S *)b->data, sa, sb, (So *)out->data, n);
    else MD_LAUNCH((k_nw_binary<F, Tc, S, MA, MB, false>), grid, MD_BLOCK, (const S *)a->data, (const S *)b->data, sa, sb, (So *)out->data, n);
    return MD_LAUNCH_CHECK("binary(narrow, stream)");
  }
  template <class F, class Tc, int DT>
  static int binary_stream_modes(const mdhip_array *a, const mdhip_array *b, Tc sa, Tc sb, const mdhip_array *out, int64_t n) {
    if (!a->is_scalar && !b->is_scalar) return binary_stream<F, Tc, DT, NM_VEC, NM_VEC>(a, b, sa, sb, out, n);
    if (!a->is_scalar) return binary_stream<F, Tc, DT, NM_VEC, NM_SCAL>(a, b, sa, sb, out, n);
    return binary_stream<F, Tc, DT, NM_SCAL, NM_VEC>(a, b, sa, sb, out, n);
  }
  // -1: the iteration space does not fit k_nw_binary_axes (the caller goes on to the generic kernel)
  template <class F, class Tc, int DT>
  static int binary_axes(const MdIter &it, const mdhip_array *a, const mdhip_array *b, Tc sa, Tc sb, const mdhip_array *out) {
    using S = typename nw_storage<DT>::type;
    using So = typename nw_out<F, Tc, S>::type;
    constexpr int E = 16 / sizeof(S);
    const int nd = it.ndim, sh = 4 - nd;
    const int64_t inner = it.shape[nd - 1];
    if ((inner % E) || it.strides[2][nd - 1] != 1 || ((uintptr_t)out->data % (sizeof(So) * E)) != 0) return -1;
    int64_t dense = inner;
    for (int d = nd - 2; d >= 0; --d) {
      if (it.strides[2][d] != dense) return -1;
      dense *= it.shape[d];
    }
    NwAxes g{};
    g.e0 = nd == 4 ? it.shape[0] : 1;
    g.e1 = nd >= 3 ? it.shape[nd - 3] : 1;
    g.e2 = it.shape[nd - 2];
    g.nv = inner / E;
    g.rows = g.e0 * g.e1 * g.e2;
    const mdhip_array *ops[2] = {a, b};
    for (int k = 0; k < 2; ++k) {
      if (ops[k]->is_scalar) continue;
      const int64_t is = it.strides[k][nd - 1];
      if (is != 0 && is != 1) return -1;
      g.in[k] = (int)is;
      for (int j = sh; j < 3; ++j) {
        const int64_t st = it.strides[k][j - sh];
        if (is == 1 && (st % E)) return -1;
        g.st[k][j] = st;
      }
      if (is == 1 && ((uintptr_t)ops[k]->data & 15)) return -1;
    }
    MD_LAUNCH((k_nw_binary_axes<F, Tc, S>), md_grid_for(g.rows * g.nv), MD_BLOCK, g, a->is_scalar ? nullptr : (const S *)a->data,
              b->is_scalar ? nullptr : (const S *)b->data, sa, sb, (So *)out->data);
    return MD_LAUNCH_CHECK("binary(narrow, axes)");
  }
  template <class F, class Tc> static int nbinary(const MdIter &it, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
    using R = decltype(F::apply(Tc(), Tc()));
    const Tc sa = a->is_scalar ? md_scalar_as<Tc>(a) : Tc(), sb = b->is_scalar ? md_scalar_as<Tc>(b) : Tc();
    const mdhip_array *ops[3] = {a, b, out};
    // stream form: the array operand(s) of ONE storage-only type, out of that type (arithmetic) or bool (comparisons), all dense
    const int sdt = !a->is_scalar ? a->dtype : b->dtype;
    const bool same = (a->is_scalar || a->dtype == sdt) && (b->is_scalar || b->dtype == sdt) && !(a->is_scalar && b->is_scalar);
    if (same && md_is_narrow(sdt) && nw_contiguous(it, 3, ops) && (md_same<R, b8>::value ? out->dtype == MDHIP_BOOL : out->dtype == sdt)) {
      switch (sdt) {
#define MD_NW_B(DT)                                                                                                                   \
  case DT:                                                                                                                            \
    if constexpr (md_same<Tc, typename md_carrier_type<DT>::type>::value) return binary_stream_modes<F, Tc, DT>(a, b, sa, sb, out, it.total); \
    break;
        MD_NW_B(MDHIP_I8) MD_NW_B(MDHIP_I16) MD_NW_B(MDHIP_U8) MD_NW_B(MDHIP_U16) MD_NW_B(MDHIP_U32) MD_NW_B(MDHIP_U64) MD_NW_B(MDHIP_F16)
#undef MD_NW_B
      }
    }
    if (same && md_is_narrow(sdt) && (md_same<R, b8>::value ? out->dtype == MDHIP_BOOL : out->dtype == sdt) && it.ndim >= 2 && it.ndim <= 4 &&
        it.total >= (1 << 14)) {
      switch (sdt) {
#define MD_NW_BA(DT)                                                                                                                  \
  case DT:                                                                                                                            \
    if constexpr (md_same<Tc, typename md_carrier_type<DT>::type>::value) {                                                           \
      int rc = binary_axes<F, Tc, DT>(it, a, b, sa, sb, out);                                                                         \
      if (rc != -1) return rc;                                                                                                        \
    }                                                                                                                                 \
    break;
        MD_NW_BA(MDHIP_I8) MD_NW_BA(MDHIP_I16) MD_NW_BA(MDHIP_U8) MD_NW_BA(MDHIP_U16) MD_NW_BA(MDHIP_U32) MD_NW_BA(MDHIP_U64) MD_NW_BA(MDHIP_F16)
#undef MD_NW_BA
      }
    }
    k_nw_binary_generic<F, Tc><<<md_grid_for(it.total), MD_BLOCK, 0, md_stream()>>>(it, a->data, a->dtype, a->is_scalar, sa, b->data, b->dtype,
                                                                                  b->is_scalar, sb, out->data, out->dtype);
    return MD_LAUNCH_CHECK("binary(narrow, generic)");
  }

  // -------------------------------------------------------------------- where ----
  template <class Tc> static int nwhere(const MdIter &it, const mdhip_array *c, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
    const uint8_t sc = c->is_scalar ? md_scalar_as<uint8_t>(c) : 0;
    const Tc sa = a->is_scalar ? md_scalar_as<Tc>(a) : Tc(), sb = b->is_scalar ? md_scalar_as<Tc>(b) : Tc();
    k_nw_where_generic<Tc><<<md_grid_for(it.total), MD_BLOCK, 0, md_stream()>>>(it, c->data, c->dtype, c->is_scalar, sc, a->data, a->dtype, a->is_scalar, sa,
                                                                                b->data, b->dtype, b->is_scalar, sb, out->data, out->dtype);
    return MD_LAUNCH_CHECK("where(narrow, generic)");
  }
};

}  // namespace

// entry points of elementwise.hip hand over here when a storage-only dtype takes part
int md_narrow_unary(int op, const mdhip_array *x, const mdhip_array *out) { return md_narrow_unary_dispatch<HipExecN>(op, x, out); }
int md_narrow_binary(int op, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out, int cdt) {
  return md_narrow_binary_dispatch<HipExecN>(op, a, b, out, cdt);
}
int md_narrow_where(const mdhip_array *c, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
  return md_narrow_where_dispatch<HipExecN>(c, a, b, out);
}
