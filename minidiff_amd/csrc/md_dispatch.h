// md_dispatch.h — (op code, dtype codes) -> template instantiation, written once
// and instantiated with an execution policy X:
//   HipExec  (csrc/*.hip)                 launches gfx950 kernels on the stream
//   HostExec (oracle/host_target/*.cpp)   plain loops, the CPU test double
// The policy supplies:
//   template<class F,class Tc,class To> static int unary (const MdIter&, const mdhip_array* x, const mdhip_array* out);
//   template<class F,class Tc,class To> static int binary(const MdIter&, const mdhip_array* a, const mdhip_array* b, const mdhip_array* out);
//   template<class T>                   static int where (const MdIter&, cond, a, b, out);
//   template<class R,class Tacc,class To> static int reduce(const MdRedPlan&, x, out);
//   template<bool IsMax,class T>        static int argreduce(const MdRedPlan&, x, out);
//   template<class T>                   static int gemm(const MdGemm&);
#pragma once
#include "md_common.h"

#define MD_FLOAT_SWITCH(dt, T, ...)                                             \
  switch (dt) {                                                                 \
    case MDHIP_F32: { using T = float; __VA_ARGS__; }                           \
    case MDHIP_F64: { using T = double; __VA_ARGS__; }                          \
    default: return md_fail(MDHIP_ETYPE, "ufunc not supported for dtype %s", md_dtype_name(dt)); \
  }
#define MD_NUM_SWITCH(dt, T, ...)                                               \
  switch (dt) {                                                                 \
    case MDHIP_I32: { using T = int32_t; __VA_ARGS__; }                         \
    case MDHIP_I64: { using T = int64_t; __VA_ARGS__; }                         \
    case MDHIP_F32: { using T = float; __VA_ARGS__; }                           \
    case MDHIP_F64: { using T = double; __VA_ARGS__; }                          \
    default: return md_fail(MDHIP_ETYPE, "ufunc not supported for dtype %s", md_dtype_name(dt)); \
  }
// bool handled as uint8 0/1
#define MD_ALL_SWITCH(dt, T, ...)                                               \
  switch (dt) {                                                                 \
    case MDHIP_BOOL: { using T = uint8_t; __VA_ARGS__; }                        \
    case MDHIP_I32: { using T = int32_t; __VA_ARGS__; }                         \
    case MDHIP_I64: { using T = int64_t; __VA_ARGS__; }                         \
    case MDHIP_F32: { using T = float; __VA_ARGS__; }                           \
    case MDHIP_F64: { using T = double; __VA_ARGS__; }                          \
    default: return md_fail(MDHIP_ETYPE, "unknown dtype code %d", (int)(dt));   \
  }

static inline int md_check_array(const mdhip_array *a, const char *what) {
  if (!a) return md_fail(MDHIP_EVALUE, "%s: null descriptor", what);
  if (a->dtype < 0 || a->dtype >= MDHIP_NUM_DTYPES) return md_fail(MDHIP_ETYPE, "%s: unknown dtype code %d", what, a->dtype);
  if (!a->is_scalar && (a->ndim < 0 || a->ndim > MDHIP_MAX_NDIM))
    return md_fail(MDHIP_EVALUE, "%s: ndim %d out of range (max %d)", what, a->ndim, MDHIP_MAX_NDIM);
  return MDHIP_OK;
}

// ================================ unary ========================================
template <class X> int md_unary_dispatch(int op, const mdhip_array *x, const mdhip_array *out) {
  MD_TRY(md_check_array(x, "unary x"));
  MD_TRY(md_check_array(out, "unary out"));
  if (out->is_scalar) return md_fail(MDHIP_EVALUE, "unary: out cannot be a scalar");
  MdIter it;
  const mdhip_array *ops[2] = {x, out};
  MD_TRY(md_build_iter(&it, 2, ops, out));
  if (it.total == 0) return MDHIP_OK;
  const int odt = out->dtype;
  switch (op) {
    case MDHIP_U_COPY:
      // astype: load in the source type, convert once to the destination type
      switch (odt) {
        case MDHIP_BOOL: return X::template unary<UCopy, b8, b8>(it, x, out);
        case MDHIP_I32: return X::template unary<UCopy, int32_t, int32_t>(it, x, out);
        case MDHIP_I64: return X::template unary<UCopy, int64_t, int64_t>(it, x, out);
        case MDHIP_F32: return X::template unary<UCopy, float, float>(it, x, out);
        default: return X::template unary<UCopy, double, double>(it, x, out);
      }
#define MD_U_NUM(code, F)                                                      \
  case code:                                                                   \
    if (odt == MDHIP_BOOL) {                                                   \
      if (code == MDHIP_U_ABS || code == MDHIP_U_CEIL || code == MDHIP_U_FLOOR) \
        return X::template unary<UCopy, b8, b8>(it, x, out);                   \
      return md_fail(MDHIP_ETYPE, "ufunc not supported for dtype bool");       \
    }                                                                          \
    MD_NUM_SWITCH(odt, T, return (X::template unary<F, T, T>(it, x, out)))
      MD_U_NUM(MDHIP_U_ABS, UAbs)
      MD_U_NUM(MDHIP_U_NEG, UNeg)
      MD_U_NUM(MDHIP_U_SIGN, USign)
      MD_U_NUM(MDHIP_U_CEIL, UCeil)
      MD_U_NUM(MDHIP_U_FLOOR, UFloor)
#undef MD_U_NUM
#define MD_U_FLT(code, F) \
  case code: MD_FLOAT_SWITCH(odt, T, return (X::template unary<F, T, T>(it, x, out)))
      MD_U_FLT(MDHIP_U_SIN, USin)
      MD_U_FLT(MDHIP_U_COS, UCos)
      MD_U_FLT(MDHIP_U_TAN, UTan)
      MD_U_FLT(MDHIP_U_SINH, USinh)
      MD_U_FLT(MDHIP_U_COSH, UCosh)
      MD_U_FLT(MDHIP_U_TANH, UTanh)
      MD_U_FLT(MDHIP_U_EXP, UExp)
      MD_U_FLT(MDHIP_U_LOG, ULog)
      MD_U_FLT(MDHIP_U_SQRT, USqrt)
#undef MD_U_FLT
    case MDHIP_U_LOGICAL_NOT:
      if (odt != MDHIP_BOOL) return md_fail(MDHIP_ETYPE, "logical_not: out must be bool");
      return X::template unary<ULogicalNot, uint8_t, b8>(it, x, out);
    case MDHIP_U_INVERT:
      if (x->dtype == MDHIP_BOOL && odt == MDHIP_BOOL) return X::template unary<ULogicalNot, uint8_t, b8>(it, x, out);
      if (x->dtype == MDHIP_I32 && odt == MDHIP_I32) return X::template unary<UInvert, int32_t, int32_t>(it, x, out);
      if (x->dtype == MDHIP_I64 && odt == MDHIP_I64) return X::template unary<UInvert, int64_t, int64_t>(it, x, out);
      return md_fail(MDHIP_ETYPE, "ufunc 'invert' not supported for the input types");
    case MDHIP_U_ISNAN:
      if (odt != MDHIP_BOOL) return md_fail(MDHIP_ETYPE, "isnan: out must be bool");
      if (x->dtype == MDHIP_F32) return X::template unary<UIsnan, float, b8>(it, x, out);
      if (x->dtype == MDHIP_F64) return X::template unary<UIsnan, double, b8>(it, x, out);
      return X::template unary<UIsnan, int64_t, b8>(it, x, out);
  }
  return md_fail(MDHIP_EVALUE, "unknown unary op code %d", op);
}

// ================================ binary =======================================
template <class X> int md_binary_dispatch(int op, const mdhip_array *a, const mdhip_array *b,
                                          const mdhip_array *out, int cdt) {
  MD_TRY(md_check_array(a, "binary a"));
  MD_TRY(md_check_array(b, "binary b"));
  MD_TRY(md_check_array(out, "binary out"));
  if (out->is_scalar) return md_fail(MDHIP_EVALUE, "binary: out cannot be a scalar");
  if (cdt < 0 || cdt >= MDHIP_NUM_DTYPES) return md_fail(MDHIP_ETYPE, "binary: bad compute dtype %d", cdt);
  MdIter it;
  const mdhip_array *ops[3] = {a, b, out};
  MD_TRY(md_build_iter(&it, 3, ops, out));
  if (it.total == 0) return MDHIP_OK;
  const int odt = out->dtype;
  const bool is_cmp = op >= MDHIP_B_EQ && op <= MDHIP_B_GE;
  const bool is_log = op >= MDHIP_B_LAND && op <= MDHIP_B_LXOR;
  if (is_cmp || is_log) {
    if (odt != MDHIP_BOOL) return md_fail(MDHIP_ETYPE, "comparison/logical ufunc writes bool, got %s", md_dtype_name(odt));
  } else if (odt != cdt) {
    return md_fail(MDHIP_ETYPE, "binary: out dtype %s != loop dtype %s", md_dtype_name(odt), md_dtype_name(cdt));
  }
  if (!is_cmp && !is_log && cdt == MDHIP_BOOL) {
    // NumPy's '?' loops: add/maximum = or, multiply/minimum = and
    switch (op) {
      case MDHIP_B_ADD: case MDHIP_B_MAXIMUM: op = MDHIP_B_LOR; break;
      case MDHIP_B_MUL: case MDHIP_B_MINIMUM: op = MDHIP_B_LAND; break;
      case MDHIP_B_SUB:
        return md_fail(MDHIP_ETYPE, "numpy boolean subtract, the `-` operator, is not supported, use the bitwise_xor, the `^` operator, or the logical_xor function instead.");
      default: return md_fail(MDHIP_ETYPE, "ufunc not supported for dtype bool");
    }
  }
  // x ** c with a host-scalar float exponent (the tape's x**2, x**1, x**0.5: definitions.py:386-391,
  // 509): one read + one write with the exponent folded into the kernel instead of a generic pow
  if (op == MDHIP_B_POW && b->is_scalar && !a->is_scalar && md_dtype_is_float(cdt) && a->dtype == cdt) {
    const double e = md_dtype_is_float(b->dtype) ? b->scalar_f : (double)b->scalar_i;
    MdIter it2;
    const mdhip_array *ops2[2] = {a, out};
    MD_TRY(md_build_iter(&it2, 2, ops2, out));
#define MD_POW_AS(F)                                                                   \
    return cdt == MDHIP_F32 ? X::template unary<F, float, float>(it2, a, out)            \
                            : X::template unary<F, double, double>(it2, a, out)
    if (e == 2.0) { MD_POW_AS(USquare); }
    if (e == 1.0) { MD_POW_AS(UCopy); }
    if (e == 0.5) { MD_POW_AS(UPowHalf); }
    if (e == -1.0) { MD_POW_AS(URecip); }
    if (e == 0.0) { MD_POW_AS(UOne); }
#undef MD_POW_AS
  }
  switch (op) {
#define MD_B_NUM(code, F) \
  case code: MD_NUM_SWITCH(cdt, T, return (X::template binary<F, T, T>(it, a, b, out)))
    MD_B_NUM(MDHIP_B_ADD, BAdd)
    MD_B_NUM(MDHIP_B_SUB, BSub)
    MD_B_NUM(MDHIP_B_MUL, BMul)
    MD_B_NUM(MDHIP_B_FLOOR_DIV, BFloorDiv)
    MD_B_NUM(MDHIP_B_MOD, BMod)
    MD_B_NUM(MDHIP_B_POW, BPow)
    MD_B_NUM(MDHIP_B_MAXIMUM, BMaximum)
    MD_B_NUM(MDHIP_B_MINIMUM, BMinimum)
#undef MD_B_NUM
    case MDHIP_B_TRUE_DIV: MD_FLOAT_SWITCH(cdt, T, return (X::template binary<BTrueDiv, T, T>(it, a, b, out)))
#define MD_B_CMP(code, F) \
  case code: MD_ALL_SWITCH(cdt, T, return (X::template binary<F, T, b8>(it, a, b, out)))
    MD_B_CMP(MDHIP_B_EQ, BEq)
    MD_B_CMP(MDHIP_B_NE, BNe)
    MD_B_CMP(MDHIP_B_LT, BLt)
    MD_B_CMP(MDHIP_B_LE, BLe)
    MD_B_CMP(MDHIP_B_GT, BGt)
    MD_B_CMP(MDHIP_B_GE, BGe)
#undef MD_B_CMP
    case MDHIP_B_LAND: return X::template binary<BLand, uint8_t, b8>(it, a, b, out);
    case MDHIP_B_LOR: return X::template binary<BLor, uint8_t, b8>(it, a, b, out);
    case MDHIP_B_LXOR: return X::template binary<BLxor, uint8_t, b8>(it, a, b, out);
  }
  return md_fail(MDHIP_EVALUE, "unknown binary op code %d", op);
}

// ================================ where ========================================
template <class X> int md_where_dispatch(const mdhip_array *cond, const mdhip_array *a, const mdhip_array *b,
                                         const mdhip_array *out) {
  MD_TRY(md_check_array(cond, "where cond"));
  MD_TRY(md_check_array(a, "where x"));
  MD_TRY(md_check_array(b, "where y"));
  MD_TRY(md_check_array(out, "where out"));
  MdIter it;
  const mdhip_array *ops[4] = {cond, a, b, out};
  MD_TRY(md_build_iter(&it, 4, ops, out));
  if (it.total == 0) return MDHIP_OK;
  switch (out->dtype) {
    case MDHIP_BOOL: return X::template where<b8>(it, cond, a, b, out);
    case MDHIP_I32: return X::template where<int32_t>(it, cond, a, b, out);
    case MDHIP_I64: return X::template where<int64_t>(it, cond, a, b, out);
    case MDHIP_F32: return X::template where<float>(it, cond, a, b, out);
    default: return X::template where<double>(it, cond, a, b, out);
  }
}

// ================================ reduce =======================================
// x may be of ANY of the twelve dtypes: the reduction kernels load through md_load (one conversion to the accumulator type,
// what NumPy's cast to the loop dtype does); `out` is one of the five compute dtypes — or uint64 for sums / products, which
// accumulate in int64 with the same bits — except for MAX / MIN of uint64, which compare as unsigned. A caller that wants a
// storage-only RESULT dtype (max of int8 -> int8, sum of float16 -> float16) reduces into the carrier type and converts the
// (small) result: mdhip_reduce does that itself.
template <class X> int md_reduce_dispatch(int op, const mdhip_array *x, const mdhip_array *out_in, uint32_t mask) {
  MD_TRY(md_check_any_array(x, "reduce x"));
  MD_TRY(md_check_any_array(out_in, "reduce out"));
  mdhip_array out_v = *out_in;
  if (out_v.dtype == MDHIP_U64 && (op == MDHIP_R_SUM || op == MDHIP_R_PROD)) out_v.dtype = MDHIP_I64;   // same bits modulo 2^64
  const mdhip_array *out = &out_v;
  const bool u64_cmp = (op == MDHIP_R_MAX || op == MDHIP_R_MIN) && x->dtype == MDHIP_U64 && out->dtype == MDHIP_U64;
  if (!u64_cmp) MD_TRY(md_check_array(out, "reduce out"));
  if (x->is_scalar || out->is_scalar) return md_fail(MDHIP_EVALUE, "reduce: scalar operands not accepted");
  MdRedPlan pl;
  MD_TRY(md_build_redplan(&pl, x, out, mask));
  if (pl.n_out == 0) return MDHIP_OK;
  const int odt = out->dtype, xdt = x->dtype;
  switch (op) {
    case MDHIP_R_SUM:
    case MDHIP_R_PROD:
      // accumulate in the output dtype (NumPy: ints widen to int64, floats keep theirs)
      switch (odt) {
#define MD_R_ACC(code, T)                                                               \
  case code:                                                                            \
    return op == MDHIP_R_SUM ? X::template reduce<RSum, T, T>(pl, x, out)               \
                             : X::template reduce<RProd, T, T>(pl, x, out);
        MD_R_ACC(MDHIP_I32, int32_t)
        MD_R_ACC(MDHIP_I64, int64_t)
        MD_R_ACC(MDHIP_F32, float)
        MD_R_ACC(MDHIP_F64, double)
#undef MD_R_ACC
        default: return md_fail(MDHIP_ETYPE, "sum/prod: unsupported accumulator dtype %s", md_dtype_name(odt));
      }
    case MDHIP_R_MAX:
    case MDHIP_R_MIN:
      if (pl.n_red == 0) return md_fail(MDHIP_EVALUE, "zero-size array to reduction operation %s which has no identity", op == MDHIP_R_MAX ? "maximum" : "minimum");
      if (u64_cmp) return op == MDHIP_R_MAX ? X::template reduce<RMax, uint64_t, uint64_t>(pl, x, out) : X::template reduce<RMin, uint64_t, uint64_t>(pl, x, out);
      // (a storage-only input compares in its carrier type: `out` then has the carrier's dtype)
      if (xdt >= MDHIP_NUM_DTYPES) {
        switch (odt) {
          case MDHIP_I32: return op == MDHIP_R_MAX ? X::template reduce<RMax, int32_t, int32_t>(pl, x, out) : X::template reduce<RMin, int32_t, int32_t>(pl, x, out);
          case MDHIP_I64: return op == MDHIP_R_MAX ? X::template reduce<RMax, int64_t, int64_t>(pl, x, out) : X::template reduce<RMin, int64_t, int64_t>(pl, x, out);
          case MDHIP_F32: return op == MDHIP_R_MAX ? X::template reduce<RMax, float, float>(pl, x, out) : X::template reduce<RMin, float, float>(pl, x, out);
        }
        return md_fail(MDHIP_ETYPE, "max/min of %s: out must have the carrier dtype", md_dtype_name(xdt));
      }
      if (odt != xdt) return md_fail(MDHIP_ETYPE, "max/min: out dtype must equal input dtype");
      MD_ALL_SWITCH(xdt, T, {
        using To = typename md_cond<md_same<T, uint8_t>::value, b8, T>::type;
        return op == MDHIP_R_MAX ? X::template reduce<RMax, T, To>(pl, x, out)
                                 : X::template reduce<RMin, T, To>(pl, x, out);
      })
    case MDHIP_R_ANY:
    case MDHIP_R_ALL:
      if (odt != MDHIP_BOOL) return md_fail(MDHIP_ETYPE, "any/all: out must be bool");
      return op == MDHIP_R_ANY ? X::template reduce<RAny, uint8_t, b8>(pl, x, out)
                               : X::template reduce<RAll, uint8_t, b8>(pl, x, out);
    case MDHIP_R_ARGMAX:
    case MDHIP_R_ARGMIN:
      if (pl.n_red == 0) return md_fail(MDHIP_EVALUE, "attempt to get %s of an empty sequence", op == MDHIP_R_ARGMAX ? "argmax" : "argmin");
      if (odt != MDHIP_I64) return md_fail(MDHIP_ETYPE, "argmax/argmin: out must be int64");
      switch (xdt) {   // storage-only inputs compare in their carrier type
        case MDHIP_I8: case MDHIP_I16: case MDHIP_U8: case MDHIP_U16:
          return op == MDHIP_R_ARGMAX ? X::template argreduce<true, int32_t>(pl, x, out) : X::template argreduce<false, int32_t>(pl, x, out);
        case MDHIP_U32:
          return op == MDHIP_R_ARGMAX ? X::template argreduce<true, int64_t>(pl, x, out) : X::template argreduce<false, int64_t>(pl, x, out);
        case MDHIP_U64:
          return op == MDHIP_R_ARGMAX ? X::template argreduce<true, uint64_t>(pl, x, out) : X::template argreduce<false, uint64_t>(pl, x, out);
        case MDHIP_F16:
          return op == MDHIP_R_ARGMAX ? X::template argreduce<true, float>(pl, x, out) : X::template argreduce<false, float>(pl, x, out);
      }
      MD_ALL_SWITCH(xdt, T, {
        return op == MDHIP_R_ARGMAX ? X::template argreduce<true, T>(pl, x, out)
                                    : X::template argreduce<false, T>(pl, x, out);
      })
  }
  return md_fail(MDHIP_EVALUE, "unknown reduce op code %d", op);
}

// ================================ matmul =======================================
template <class X> int md_matmul_dispatch(const mdhip_array *a, const mdhip_array *b, const mdhip_array *c) {
  MD_TRY(md_check_array(a, "matmul a"));
  MD_TRY(md_check_array(b, "matmul b"));
  MD_TRY(md_check_array(c, "matmul c"));
  MdGemm g{};
  MD_TRY(md_build_gemm(&g, a, b, c));
  if (g.batch == 0 || g.M == 0 || g.N == 0) return MDHIP_OK;
  switch (c->dtype) {
    case MDHIP_F32: return X::template gemm<float>(g);
    case MDHIP_F64: return X::template gemm<double>(g);
    case MDHIP_I32: return X::template gemm<int32_t>(g);
    case MDHIP_I64: return X::template gemm<int64_t>(g);
  }
  return md_fail(MDHIP_ETYPE, "matmul not supported for dtype %s", md_dtype_name(c->dtype));
}

// ================================ index plans ===================================
static inline int md_check_plan(const mdhip_index_plan *pl) {
  if (!pl) return md_fail(MDHIP_EVALUE, "index plan is null");
  if (pl->ndim < 0 || pl->ndim > MDHIP_MAX_NDIM) return md_fail(MDHIP_EVALUE, "index plan ndim %d out of range", pl->ndim);
  if (pl->n_idx < 0 || pl->n_idx > MDHIP_MAX_NDIM) return md_fail(MDHIP_EVALUE, "index plan n_idx %d out of range", pl->n_idx);
  for (int d = 0; d < pl->ndim; ++d)
    if (pl->shape[d] < 0) return md_fail(MDHIP_EVALUE, "index plan extent %lld of dimension %d is negative", (long long)pl->shape[d], d);
  for (int k = 0; k < pl->n_idx; ++k) {
    if (pl->idx_extent[k] < 0) return md_fail(MDHIP_EVALUE, "index plan: indexed axis %d has a negative extent", k);
    if (pl->idx_dtype[k] != MDHIP_I32 && pl->idx_dtype[k] != MDHIP_I64)
      return md_fail(MDHIP_EINDEX, "arrays used as indices must be of integer (or boolean) type");
    if (!pl->idx_ptr[k]) return md_fail(MDHIP_EVALUE, "index array %d is null", k);
  }
  return MDHIP_OK;
}
static inline int64_t md_plan_total(const mdhip_index_plan *pl) {
  int64_t n = 1;
  for (int d = 0; d < pl->ndim; ++d) n *= pl->shape[d];
  return n;
}
// offset(p) of the plan; *oob set when an index is out of bounds
MD_HD int64_t md_plan_offset(const mdhip_index_plan &pl, int64_t lin, int64_t *pos, bool *oob) {
  int64_t off = 0;
  for (int d = pl.ndim - 1; d >= 0; --d) {
    int64_t e = pl.shape[d];
    int64_t q = lin / e, r = lin - q * e;
    lin = q;
    pos[d] = r;
    off += r * pl.src_strides[d];
  }
  for (int k = 0; k < pl.n_idx; ++k) {
    int64_t io = 0;
    for (int d = 0; d < pl.ndim; ++d) io += pos[d] * pl.idx_strides[k][d];
    int64_t v = pl.idx_dtype[k] == MDHIP_I64 ? ((const int64_t *)pl.idx_ptr[k])[io]
                                             : (int64_t)((const int32_t *)pl.idx_ptr[k])[io];
    int64_t ext = pl.idx_extent[k];
    if (v < -ext || v >= ext) { *oob = true; v = 0; }
    if (v < 0) v += ext;
    off += v * pl.idx_mult[k];
  }
  return off;
}
