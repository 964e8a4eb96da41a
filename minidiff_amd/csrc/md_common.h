// md_common.h — plumbing shared by the gfx950 library and the CPU test double:
// error reporting, dtype helpers, and the iteration-space builders that turn the
// caller's (shape, strides) descriptors into the collapsed forms the kernels walk.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/mdhip.h"
#include "md_ops.h"
#include "md_options.h"

// ---- errors ------------------------------------------------------------------
std::string &md_err_slot();  // thread-local storage lives in the runtime TU
static inline int md_fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  md_err_slot() = buf;
  return code;
}
#define MD_TRY(expr)            \
  do {                          \
    int _st = (expr);           \
    if (_st != MDHIP_OK) return _st; \
  } while (0)

static inline size_t md_dtype_size(int dt) {
  switch (dt) {
    case MDHIP_BOOL: return 1;
    case MDHIP_I32: return 4;
    case MDHIP_I64: return 8;
    case MDHIP_F32: return 4;
    case MDHIP_F64: return 8;
    case MDHIP_I8: case MDHIP_U8: return 1;
    case MDHIP_I16: case MDHIP_U16: case MDHIP_F16: return 2;
    case MDHIP_U32: return 4;
    case MDHIP_U64: return 8;
  }
  return 0;
}
static inline const char *md_dtype_name(int dt) {
  switch (dt) {
    case MDHIP_BOOL: return "bool";
    case MDHIP_I32: return "int32";
    case MDHIP_I64: return "int64";
    case MDHIP_F32: return "float32";
    case MDHIP_F64: return "float64";
    case MDHIP_I8: return "int8";
    case MDHIP_I16: return "int16";
    case MDHIP_U8: return "uint8";
    case MDHIP_U16: return "uint16";
    case MDHIP_U32: return "uint32";
    case MDHIP_U64: return "uint64";
    case MDHIP_F16: return "float16";
  }
  return "?";
}
static inline bool md_dtype_is_float(int dt) { return dt == MDHIP_F32 || dt == MDHIP_F64; }
static inline bool md_dtype_is_unsigned(int dt) { return dt == MDHIP_U8 || dt == MDHIP_U16 || dt == MDHIP_U32 || dt == MDHIP_U64; }

template <class T> struct md_dtype_of;
template <> struct md_dtype_of<b8> { static constexpr int value = MDHIP_BOOL; };
template <> struct md_dtype_of<uint8_t> { static constexpr int value = MDHIP_BOOL; };
template <> struct md_dtype_of<int32_t> { static constexpr int value = MDHIP_I32; };
template <> struct md_dtype_of<int64_t> { static constexpr int value = MDHIP_I64; };
template <> struct md_dtype_of<uint64_t> { static constexpr int value = MDHIP_U64; };
template <> struct md_dtype_of<float> { static constexpr int value = MDHIP_F32; };
template <> struct md_dtype_of<double> { static constexpr int value = MDHIP_F64; };

// ---- typed element access with a runtime source dtype ------------------------
// Tc == uint8_t means "truth value": any source dtype loads as (x != 0).
// (the seven storage-only dtypes of include/mdhip.h load like the five compute dtypes: one conversion to the compute type —
// what NumPy's cast of an operand to the loop dtype does; the switch is wave-uniform)
template <class Tc> MD_HD Tc md_load(const void *p, int dtype, int64_t off) {
  if constexpr (md_same<Tc, uint8_t>::value) {
    switch (dtype) {
      case MDHIP_BOOL: case MDHIP_I8: case MDHIP_U8: return (uint8_t)(((const uint8_t *)p)[off] != 0);
      case MDHIP_I16: case MDHIP_U16: return (uint8_t)(((const uint16_t *)p)[off] != 0);
      case MDHIP_F16: return (uint8_t)((((const uint16_t *)p)[off] & 0x7FFFu) != 0);   // (-0.0 is false, NaN is true)
      case MDHIP_I32: case MDHIP_U32: return (uint8_t)(((const int32_t *)p)[off] != 0);
      case MDHIP_I64: case MDHIP_U64: return (uint8_t)(((const int64_t *)p)[off] != 0);
      case MDHIP_F32: return (uint8_t)(((const float *)p)[off] != 0.0f);
      default: return (uint8_t)(((const double *)p)[off] != 0.0);
    }
  } else {
    switch (dtype) {
      case MDHIP_BOOL: return md_cast<Tc>(((const b8 *)p)[off]);
      case MDHIP_I32: return md_cast<Tc>(((const int32_t *)p)[off]);
      case MDHIP_I64: return md_cast<Tc>(((const int64_t *)p)[off]);
      case MDHIP_F32: return md_cast<Tc>(((const float *)p)[off]);
      case MDHIP_I8: return md_cast<Tc>(((const int8_t *)p)[off]);
      case MDHIP_I16: return md_cast<Tc>(((const int16_t *)p)[off]);
      case MDHIP_U8: return md_cast<Tc>(((const uint8_t *)p)[off]);
      case MDHIP_U16: return md_cast<Tc>(((const uint16_t *)p)[off]);
      case MDHIP_U32: return md_cast<Tc>(((const uint32_t *)p)[off]);
      case MDHIP_U64: return md_cast<Tc>(((const uint64_t *)p)[off]);
      case MDHIP_F16: return md_cast<Tc>(((const f16 *)p)[off]);
      default: return md_cast<Tc>(((const double *)p)[off]);
    }
  }
}
// store a result of the compute type (or a truth value b8) into an array of ANY of the twelve dtypes: integer destinations
// truncate (two's-complement wrap-around — the narrow result of an operation carried out in a wider integer), floats round once
template <class R> MD_HD void md_store_as(void *p, int dtype, int64_t off, R v) {
  if constexpr (md_same<R, b8>::value) {
    switch (dtype) {
      case MDHIP_BOOL: case MDHIP_I8: case MDHIP_U8: ((uint8_t *)p)[off] = v.v; return;
      case MDHIP_I16: case MDHIP_U16: ((uint16_t *)p)[off] = v.v; return;
      case MDHIP_I32: case MDHIP_U32: ((uint32_t *)p)[off] = v.v; return;
      case MDHIP_I64: case MDHIP_U64: ((uint64_t *)p)[off] = v.v; return;
      case MDHIP_F16: ((f16 *)p)[off] = md_cast<f16>((float)v.v); return;
      case MDHIP_F32: ((float *)p)[off] = (float)v.v; return;
      default: ((double *)p)[off] = (double)v.v; return;
    }
  } else {
    switch (dtype) {
      case MDHIP_BOOL: ((uint8_t *)p)[off] = (uint8_t)(v != (R)0); return;
      case MDHIP_I8: case MDHIP_U8: ((uint8_t *)p)[off] = (uint8_t)md_cast<int64_t>(v); return;
      case MDHIP_I16: case MDHIP_U16: ((uint16_t *)p)[off] = (uint16_t)md_cast<int64_t>(v); return;
      case MDHIP_I32: case MDHIP_U32: ((uint32_t *)p)[off] = (uint32_t)md_cast<int64_t>(v); return;
      case MDHIP_I64: ((int64_t *)p)[off] = md_cast<int64_t>(v); return;
      case MDHIP_U64: ((uint64_t *)p)[off] = md_cast<uint64_t>(v); return;
      case MDHIP_F16: ((f16 *)p)[off] = md_cast<f16>(v); return;
      case MDHIP_F32: ((float *)p)[off] = (float)v; return;
      default: ((double *)p)[off] = (double)v; return;
    }
  }
}
// result of a functor (T or b8) -> storage type To
template <class To, class R> MD_HD To md_to_out(R r) { return md_cast<To>(r); }

// scalar operand -> compute type
// (a scalar descriptor of dtype MDHIP_U64 carries the BITS of a value >= 2**63 in scalar_i)
template <class Tc> static inline Tc md_scalar_as(const mdhip_array *s) {
  if constexpr (md_same<Tc, uint8_t>::value) {
    return md_dtype_is_float(s->dtype) ? (uint8_t)(s->scalar_f != 0.0) : (uint8_t)(s->scalar_i != 0);
  } else {
    if (s->dtype == MDHIP_U64) return md_cast<Tc>((uint64_t)s->scalar_i);
    return md_dtype_is_float(s->dtype) ? md_cast<Tc>(s->scalar_f) : md_cast<Tc>(s->scalar_i);
  }
}

// ---- storage-only dtypes: conversion through a carrier (mdhip_convert) ---------------
// (binary16 <-> double by bit manipulation: md_half_to_double / md_double_to_half, md_ops.h)
// carrier of a dtype: 0 = int64 (signed ints, bool), 1 = uint64 (unsigned ints), 2 = double (floats)
static inline int md_dtype_carrier(int dt) {
  switch (dt) {
    case MDHIP_F32: case MDHIP_F64: case MDHIP_F16: return 2;
    case MDHIP_U8: case MDHIP_U16: case MDHIP_U32: case MDHIP_U64: return 1;
  }
  return 0;
}
template <class C> MD_HD C md_load_any(const void *p, int dt, int64_t off) {
  switch (dt) {
    case MDHIP_BOOL: return (C)(((const uint8_t *)p)[off] != 0);
    case MDHIP_I8: return (C)((const int8_t *)p)[off];
    case MDHIP_I16: return (C)((const int16_t *)p)[off];
    case MDHIP_I32: return (C)((const int32_t *)p)[off];
    case MDHIP_I64: return (C)((const int64_t *)p)[off];
    case MDHIP_U8: return (C)((const uint8_t *)p)[off];
    case MDHIP_U16: return (C)((const uint16_t *)p)[off];
    case MDHIP_U32: return (C)((const uint32_t *)p)[off];
    case MDHIP_U64: return (C)((const uint64_t *)p)[off];
    case MDHIP_F16: return (C)md_half_to_double(((const uint16_t *)p)[off]);
    case MDHIP_F32: return (C)((const float *)p)[off];
  }
  return (C)((const double *)p)[off];
}
template <class C> MD_HD int64_t md_carrier_to_i64(C v) {
  if constexpr (sizeof(C) == 8 && !(C(0.5) > C(0))) return (int64_t)v;      // integer carriers: same bits
  else return (v >= (C)9223372036854775808.0) ? (int64_t)(uint64_t)v : (int64_t)v;   // double: values of the upper unsigned half keep their bits
}
template <class C> MD_HD void md_store_any(void *p, int dt, int64_t off, C v) {
  constexpr bool is_float = C(0.5) > C(0);
  switch (dt) {
    case MDHIP_BOOL: ((uint8_t *)p)[off] = (uint8_t)(v != (C)0); return;
    case MDHIP_F16: ((uint16_t *)p)[off] = md_double_to_half((double)v); return;
    case MDHIP_F32: ((float *)p)[off] = (float)v; return;
    case MDHIP_F64: ((double *)p)[off] = (double)v; return;
  }
  // integer destinations: through 64 bits, then truncation (two's complement wrap-around, as C / NumPy's unsafe cast)
  int64_t i;
  if constexpr (is_float) i = md_carrier_to_i64(v);
  else i = (int64_t)v;
  switch (dt) {
    case MDHIP_I8: case MDHIP_U8: ((uint8_t *)p)[off] = (uint8_t)i; return;
    case MDHIP_I16: case MDHIP_U16: ((uint16_t *)p)[off] = (uint16_t)i; return;
    case MDHIP_I32: case MDHIP_U32: ((uint32_t *)p)[off] = (uint32_t)i; return;
    default: ((uint64_t *)p)[off] = (uint64_t)i; return;
  }
}
// checks + iteration space of mdhip_convert, shared by the library and the test double
static inline int md_check_any_array(const mdhip_array *a, const char *what) {
  if (!a) return md_fail(MDHIP_EVALUE, "%s: null descriptor", what);
  if (a->dtype < 0 || a->dtype >= MDHIP_NUM_ALL_DTYPES) return md_fail(MDHIP_ETYPE, "%s: unknown dtype code %d", what, a->dtype);
  if (a->is_scalar) return md_fail(MDHIP_EVALUE, "%s: scalar descriptors are not accepted", what);
  if (a->ndim < 0 || a->ndim > MDHIP_MAX_NDIM) return md_fail(MDHIP_EVALUE, "%s: ndim %d out of range (max %d)", what, a->ndim, MDHIP_MAX_NDIM);
  return MDHIP_OK;
}

// ---- collapsed N-operand iteration space --------------------------------------
#define MD_MAX_OPS 4
struct MdIter {
  int32_t ndim;
  int32_t nops;
  int64_t total;
  int64_t shape[MDHIP_MAX_NDIM];
  int64_t strides[MD_MAX_OPS][MDHIP_MAX_NDIM];  // scalar operands: all zero
};

// All operands share ndim/shape (the caller has broadcast them). Drops extent-1
// axes and merges neighbours that are jointly contiguous for every operand, so
// the common cases come out 1-D (fully contiguous) or 2-D (row broadcast).
static inline int md_build_iter(MdIter *it, int nops, const mdhip_array *const *ops,
                                const mdhip_array *shape_from) {
  int nd = shape_from->ndim;
  if (nd < 0 || nd > MDHIP_MAX_NDIM) return md_fail(MDHIP_EVALUE, "ndim %d out of range", nd);
  for (int k = 0; k < nops; ++k) {
    if (ops[k]->is_scalar) continue;
    if (ops[k]->ndim != nd) return md_fail(MDHIP_EVALUE, "operand %d ndim %d != %d", k, ops[k]->ndim, nd);
    for (int d = 0; d < nd; ++d)
      if (ops[k]->shape[d] != shape_from->shape[d])
        return md_fail(MDHIP_EVALUE, "operand %d shape mismatch on axis %d (%lld vs %lld)", k, d,
                       (long long)ops[k]->shape[d], (long long)shape_from->shape[d]);
    if (ops[k]->data == nullptr && nd >= 0) {
      int64_t n = 1;
      for (int d = 0; d < nd; ++d) n *= ops[k]->shape[d];
      if (n != 0) return md_fail(MDHIP_EVALUE, "operand %d has a null data pointer", k);
    }
  }
  it->nops = nops;
  it->total = 1;
  int64_t shp[MDHIP_MAX_NDIM];
  int64_t str[MD_MAX_OPS][MDHIP_MAX_NDIM];
  int m = 0;
  for (int d = 0; d < nd; ++d) {
    int64_t e = shape_from->shape[d];
    if (e < 0) return md_fail(MDHIP_EVALUE, "negative extent");
    it->total *= e;
    if (e == 1) continue;
    shp[m] = e;
    for (int k = 0; k < nops; ++k) str[k][m] = ops[k]->is_scalar ? 0 : ops[k]->strides[d];
    ++m;
  }
  // merge from the inside out
  int w = 0;
  for (int d = 0; d < m; ++d) {
    if (w > 0) {
      bool ok = true;
      for (int k = 0; k < nops; ++k)
        if (str[k][w - 1] != str[k][d] * shp[d]) { ok = false; break; }
      if (ok) {
        shp[w - 1] *= shp[d];
        for (int k = 0; k < nops; ++k) str[k][w - 1] = str[k][d];
        continue;
      }
    }
    shp[w] = shp[d];
    for (int k = 0; k < nops; ++k) str[k][w] = str[k][d];
    ++w;
  }
  it->ndim = w;
  for (int d = 0; d < w; ++d) {
    it->shape[d] = shp[d];
    for (int k = 0; k < nops; ++k) it->strides[k][d] = str[k][d];
  }
  for (int d = w; d < MDHIP_MAX_NDIM; ++d) {
    it->shape[d] = 1;
    for (int k = 0; k < MD_MAX_OPS; ++k) it->strides[k][d] = 0;
  }
  return MDHIP_OK;
}

// offset of operand k at linear position `lin` of the collapsed space
MD_HD void md_iter_offsets(const MdIter &it, int64_t lin, int64_t *offs) {
  for (int k = 0; k < it.nops; ++k) offs[k] = 0;
  for (int d = it.ndim - 1; d >= 0; --d) {
    int64_t e = it.shape[d];
    int64_t q = lin / e;
    int64_t r = lin - q * e;
    lin = q;
    for (int k = 0; k < it.nops; ++k) offs[k] += r * it.strides[k][d];
  }
}

// ---- reduction plan -------------------------------------------------------------
// x is split into kept axes (indexed by the output) and reduced axes; both lists
// are collapsed independently. out strides follow the kept axes.
struct MdRedPlan {
  int32_t nk, nr;
  int64_t n_out, n_red;
  int64_t kshape[MDHIP_MAX_NDIM], kx[MDHIP_MAX_NDIM], ko[MDHIP_MAX_NDIM];
  int64_t rshape[MDHIP_MAX_NDIM], rx[MDHIP_MAX_NDIM];
};
static inline int md_build_redplan(MdRedPlan *pl, const mdhip_array *x, const mdhip_array *out,
                                   uint32_t mask) {
  int nd = x->ndim;
  if (nd < 0 || nd > MDHIP_MAX_NDIM) return md_fail(MDHIP_EVALUE, "ndim %d out of range", nd);
  if (out->ndim != nd) return md_fail(MDHIP_EVALUE, "reduce: out ndim %d != x ndim %d", out->ndim, nd);
  pl->nk = pl->nr = 0;
  pl->n_out = pl->n_red = 1;
  for (int d = 0; d < nd; ++d) {
    int64_t e = x->shape[d];
    bool red = (mask >> d) & 1u;
    if (red) {
      if (out->shape[d] != 1) return md_fail(MDHIP_EVALUE, "reduce: out extent on reduced axis %d must be 1", d);
      pl->n_red *= e;
      if (e == 1) continue;
      int j = pl->nr;
      if (j > 0 && pl->rx[j - 1] == x->strides[d] * e) {
        pl->rshape[j - 1] *= e;
        pl->rx[j - 1] = x->strides[d];
      } else {
        pl->rshape[j] = e;
        pl->rx[j] = x->strides[d];
        pl->nr++;
      }
    } else {
      if (out->shape[d] != e) return md_fail(MDHIP_EVALUE, "reduce: out extent mismatch on kept axis %d", d);
      pl->n_out *= e;
      if (e == 1) continue;
      int j = pl->nk;
      if (j > 0 && pl->kx[j - 1] == x->strides[d] * e && pl->ko[j - 1] == out->strides[d] * e) {
        pl->kshape[j - 1] *= e;
        pl->kx[j - 1] = x->strides[d];
        pl->ko[j - 1] = out->strides[d];
      } else {
        pl->kshape[j] = e;
        pl->kx[j] = x->strides[d];
        pl->ko[j] = out->strides[d];
        pl->nk++;
      }
    }
  }
  for (int d = pl->nk; d < MDHIP_MAX_NDIM; ++d) { pl->kshape[d] = 1; pl->kx[d] = 0; pl->ko[d] = 0; }
  for (int d = pl->nr; d < MDHIP_MAX_NDIM; ++d) { pl->rshape[d] = 1; pl->rx[d] = 0; }
  return MDHIP_OK;
}
MD_HD void md_red_kept_offsets(const MdRedPlan &pl, int64_t o, int64_t *xoff, int64_t *ooff) {
  int64_t xo = 0, oo = 0;
  for (int d = pl.nk - 1; d >= 0; --d) {
    int64_t e = pl.kshape[d];
    int64_t q = o / e, r = o - q * e;
    o = q;
    xo += r * pl.kx[d];
    oo += r * pl.ko[d];
  }
  *xoff = xo;
  *ooff = oo;
}
MD_HD int64_t md_red_offset(const MdRedPlan &pl, int64_t r) {
  int64_t off = 0;
  for (int d = pl.nr - 1; d >= 0; --d) {
    int64_t e = pl.rshape[d];
    int64_t q = r / e, m = r - q * e;
    r = q;
    off += m * pl.rx[d];
  }
  return off;
}

// ---- matmul descriptor --------------------------------------------------------
struct MdGemm {
  int64_t batch, M, N, K;
  const void *a, *b;
  void *c;
  int64_t a_bs, a_ms, a_ks;  // element strides
  int64_t b_bs, b_ks, b_ns;
  int64_t c_bs, c_ms, c_ns;
};
static inline int md_build_gemm(MdGemm *g, const mdhip_array *a, const mdhip_array *b, const mdhip_array *c) {
  if (a->ndim != 3 || b->ndim != 3 || c->ndim != 3)
    return md_fail(MDHIP_EVALUE, "matmul: operands must be passed 3-D (batch, rows, cols)");
  if (a->dtype != b->dtype || a->dtype != c->dtype)
    return md_fail(MDHIP_ETYPE, "matmul: dtypes must agree (%s, %s -> %s)", md_dtype_name(a->dtype),
                   md_dtype_name(b->dtype), md_dtype_name(c->dtype));
  g->batch = c->shape[0];
  g->M = c->shape[1];
  g->N = c->shape[2];
  g->K = a->shape[2];
  if (a->shape[1] != g->M || b->shape[1] != g->K || b->shape[2] != g->N)
    return md_fail(MDHIP_EVALUE,
                   "matmul: Input operand 1 has a mismatch in its core dimension 0 (A is %lldx%lld, B is %lldx%lld)",
                   (long long)a->shape[1], (long long)a->shape[2], (long long)b->shape[1], (long long)b->shape[2]);
  if ((a->shape[0] != g->batch && a->shape[0] != 1) || (b->shape[0] != g->batch && b->shape[0] != 1))
    return md_fail(MDHIP_EVALUE, "matmul: batch extents do not broadcast");
  g->a = a->data; g->b = b->data; g->c = c->data;
  g->a_bs = a->shape[0] == 1 ? 0 : a->strides[0]; g->a_ms = a->strides[1]; g->a_ks = a->strides[2];
  g->b_bs = b->shape[0] == 1 ? 0 : b->strides[0]; g->b_ks = b->strides[1]; g->b_ns = b->strides[2];
  g->c_bs = c->strides[0]; g->c_ms = c->strides[1]; g->c_ns = c->strides[2];
  // (B, M, K) @ (K, N) with the batch of `a` and of `c` laid out as more rows: ONE product of B * M rows (a linear layer applied to a
  // 3-D input; B products of M rows each fill the chip worse, M < tile height worst of all)
  if (g->batch > 1 && g->b_bs == 0 && g->M > 0 && g->a_bs == g->M * g->a_ms && g->c_bs == g->M * g->c_ms) {
    g->M *= g->batch;
    g->batch = 1;
    g->a_bs = g->c_bs = 0;
  }
  return MDHIP_OK;
}
