// md_narrow.h — elementwise dispatch for calls that involve a STORAGE-ONLY dtype (include/mdhip.h: int8/16, uint8/16/32/64,
// float16 — the names of the reference table minidiff/backend/numpy.py:188-200 beyond the five the tuned kernels compute in).
//
// NumPy runs such a call in the loop dtype `cdt` it resolves (int8 * int8 -> the 'bb->b' loop: wraps in 8 bits; float16 loops
// compute in float32 and round once; int8 / int8 -> the 'dd->d' loop after casting both operands). Here: ONE launch that loads
// every operand in its own storage type, converts it to the CARRIER of the loop dtype, applies the same functor as the wide
// kernels (md_ops.h) and stores in the result's storage type —
//     int8, int16, uint8, uint16 (and int32)  -> int32 arithmetic, truncated on the store (the low bits of +, -, *, ** are the
//                                                narrow result; //, %, comparisons, max / min see the true values)
//     uint32 (and int64)                      -> int64
//     uint64                                  -> uint64 (its own unsigned loops: values >= 2**63 included)
//     float16 (and float32)                   -> float32, rounded to nearest-even on the store
//     float64                                 -> float64
// so the traffic is the algorithmic one (1.0x) instead of promote -> wide kernel -> demote (three launches, >= 3x the bytes).
// Written once, instantiated with an execution policy X (narrow.hip: gfx950 kernels; oracle/host_target: plain loops):
//   template<class F,class Tc> static int nunary (const MdIter&, const mdhip_array* x, const mdhip_array* out);
//   template<class F,class Tc> static int nbinary(const MdIter&, const mdhip_array* a, const mdhip_array* b, const mdhip_array* out);
//   template<class Tc>         static int nwhere (const MdIter&, cond, a, b, out);
#pragma once
#include "md_dispatch.h"

static inline bool md_is_narrow(int dt) { return dt >= MDHIP_NUM_DTYPES && dt < MDHIP_NUM_ALL_DTYPES; }
static inline bool md_dtype_is_int(int dt) {
  switch (dt) {
    case MDHIP_I8: case MDHIP_I16: case MDHIP_I32: case MDHIP_I64: case MDHIP_U8: case MDHIP_U16: case MDHIP_U32: case MDHIP_U64: return true;
  }
  return false;
}
static inline bool md_dtype_is_anyfloat(int dt) { return dt == MDHIP_F16 || dt == MDHIP_F32 || dt == MDHIP_F64; }

// the compute type ("carrier") of a loop dtype, as a type
template <int DT> struct md_carrier_type { using type = int32_t; };   // int8, int16, uint8, uint16, int32
template <> struct md_carrier_type<MDHIP_BOOL> { using type = uint8_t; };
template <> struct md_carrier_type<MDHIP_U32> { using type = int64_t; };
template <> struct md_carrier_type<MDHIP_I64> { using type = int64_t; };
template <> struct md_carrier_type<MDHIP_U64> { using type = uint64_t; };
template <> struct md_carrier_type<MDHIP_F16> { using type = float; };
template <> struct md_carrier_type<MDHIP_F32> { using type = float; };
template <> struct md_carrier_type<MDHIP_F64> { using type = double; };

// carrier switches: T = the compute type of loop dtype `dt`
#define MD_CARRIER_INT_CASES(T, ...)                                                              \
    case MDHIP_I8: case MDHIP_I16: case MDHIP_U8: case MDHIP_U16: case MDHIP_I32: { using T = int32_t; __VA_ARGS__; } \
    case MDHIP_U32: case MDHIP_I64: { using T = int64_t; __VA_ARGS__; }                           \
    case MDHIP_U64: { using T = uint64_t; __VA_ARGS__; }
#define MD_CARRIER_FLOAT_CASES(T, ...)                                                            \
    case MDHIP_F16: case MDHIP_F32: { using T = float; __VA_ARGS__; }                             \
    case MDHIP_F64: { using T = double; __VA_ARGS__; }
#define MD_CARRIER_NUM_SWITCH(dt, T, ...)                                                         \
  switch (dt) {                                                                                   \
    MD_CARRIER_INT_CASES(T, __VA_ARGS__)                                                          \
    MD_CARRIER_FLOAT_CASES(T, __VA_ARGS__)                                                        \
    default: return md_fail(MDHIP_ETYPE, "ufunc not supported for dtype %s", md_dtype_name(dt));  \
  }
#define MD_CARRIER_INT_SWITCH(dt, T, ...)                                                         \
  switch (dt) {                                                                                   \
    MD_CARRIER_INT_CASES(T, __VA_ARGS__)                                                          \
    default: return md_fail(MDHIP_ETYPE, "ufunc not supported for dtype %s", md_dtype_name(dt));  \
  }
#define MD_CARRIER_FLOAT_SWITCH(dt, T, ...)                                                       \
  switch (dt) {                                                                                   \
    MD_CARRIER_FLOAT_CASES(T, __VA_ARGS__)                                                        \
    default: return md_fail(MDHIP_ETYPE, "ufunc not supported for dtype %s", md_dtype_name(dt));  \
  }
#define MD_CARRIER_ALL_SWITCH(dt, T, ...)                                                         \
  switch (dt) {                                                                                   \
    case MDHIP_BOOL: { using T = uint8_t; __VA_ARGS__; }                                          \
    MD_CARRIER_INT_CASES(T, __VA_ARGS__)                                                          \
    MD_CARRIER_FLOAT_CASES(T, __VA_ARGS__)                                                        \
    default: return md_fail(MDHIP_ETYPE, "unknown dtype code %d", (int)(dt));                     \
  }

// A weak Python scalar of a float16 loop is a float16 VALUE in NumPy (np.float16(1.5) + 0.1 adds 0.0999755859375): the scalar
// descriptor is rounded to the loop dtype before it meets the float32 arithmetic.
static inline mdhip_array md_scalar_in_loop_dtype(const mdhip_array *s, int cdt) {
  mdhip_array r = *s;
  if (s->is_scalar && cdt == MDHIP_F16 && md_dtype_is_float(s->dtype)) r.scalar_f = md_half_to_double(md_double_to_half(s->scalar_f));
  if (s->is_scalar && cdt == MDHIP_F16 && (s->dtype == MDHIP_I64 || s->dtype == MDHIP_U64 || s->dtype == MDHIP_I32)) {
    // .. a Python int too: float16(70000) is inf before the arithmetic sees it
    const double v = s->dtype == MDHIP_U64 ? (double)(uint64_t)s->scalar_i : (double)s->scalar_i;
    r.dtype = MDHIP_F64;
    r.scalar_f = md_half_to_double(md_double_to_half(v));
  }
  if (s->is_scalar && cdt == MDHIP_F32 && md_dtype_is_float(s->dtype)) r.scalar_f = (double)(float)s->scalar_f;
  return r;
}

// ================================ unary ========================================
// The loop dtype of a unary call is the output's (comparison-like ones: the input's).
template <class X> int md_narrow_unary_dispatch(int op, const mdhip_array *x, const mdhip_array *out) {
  MD_TRY(md_check_any_array(out, "unary out"));
  if (!x->is_scalar) MD_TRY(md_check_any_array(x, "unary x"));
  MdIter it;
  const mdhip_array *ops[2] = {x, out};
  MD_TRY(md_build_iter(&it, 2, ops, out));
  if (it.total == 0) return MDHIP_OK;
  const int odt = out->dtype, xdt = x->dtype;
  switch (op) {
    case MDHIP_U_ABS:
      if (odt == MDHIP_BOOL) break;
      MD_CARRIER_NUM_SWITCH(odt, T, return (X::template nunary<UAbs, T>(it, x, out)))
    case MDHIP_U_NEG: MD_CARRIER_NUM_SWITCH(odt, T, return (X::template nunary<UNeg, T>(it, x, out)))
    case MDHIP_U_SIGN: MD_CARRIER_NUM_SWITCH(odt, T, return (X::template nunary<USign, T>(it, x, out)))
    case MDHIP_U_CEIL:
      if (md_dtype_is_int(odt)) { MD_CARRIER_INT_SWITCH(odt, T, return (X::template nunary<UCopy, T>(it, x, out))) }
      MD_CARRIER_FLOAT_SWITCH(odt, T, return (X::template nunary<UCeil, T>(it, x, out)))
    case MDHIP_U_FLOOR:
      if (md_dtype_is_int(odt)) { MD_CARRIER_INT_SWITCH(odt, T, return (X::template nunary<UCopy, T>(it, x, out))) }
      MD_CARRIER_FLOAT_SWITCH(odt, T, return (X::template nunary<UFloor, T>(it, x, out)))
#define MD_NU_FLT(code, F) \
  case code: MD_CARRIER_FLOAT_SWITCH(odt, T, return (X::template nunary<F, T>(it, x, out)))
      MD_NU_FLT(MDHIP_U_SIN, USin)
      MD_NU_FLT(MDHIP_U_COS, UCos)
      MD_NU_FLT(MDHIP_U_TAN, UTan)
      MD_NU_FLT(MDHIP_U_SINH, USinh)
      MD_NU_FLT(MDHIP_U_COSH, UCosh)
      MD_NU_FLT(MDHIP_U_TANH, UTanh)
      MD_NU_FLT(MDHIP_U_EXP, UExp)
      MD_NU_FLT(MDHIP_U_LOG, ULog)
      MD_NU_FLT(MDHIP_U_SQRT, USqrt)
#undef MD_NU_FLT
    case MDHIP_U_LOGICAL_NOT:
      if (odt != MDHIP_BOOL) return md_fail(MDHIP_ETYPE, "logical_not: out must be bool");
      return X::template nunary<ULogicalNot, uint8_t>(it, x, out);
    case MDHIP_U_INVERT:
      if (odt == MDHIP_BOOL && xdt == MDHIP_BOOL) return X::template nunary<ULogicalNot, uint8_t>(it, x, out);
      if (!md_dtype_is_int(odt)) return md_fail(MDHIP_ETYPE, "ufunc 'invert' not supported for the input types");
      MD_CARRIER_INT_SWITCH(odt, T, return (X::template nunary<UInvert, T>(it, x, out)))
    case MDHIP_U_ISNAN:
      if (odt != MDHIP_BOOL) return md_fail(MDHIP_ETYPE, "isnan: out must be bool");
      MD_CARRIER_NUM_SWITCH(xdt, T, return (X::template nunary<UIsnan, T>(it, x, out)))
    case MDHIP_U_COPY: break;   // conversions are mdhip_convert's
  }
  return md_fail(MDHIP_ETYPE, "unary op %d is not defined for (%s -> %s)", op, md_dtype_name(xdt), md_dtype_name(odt));
}

// ================================ binary =======================================
template <class X> int md_narrow_binary_dispatch(int op, const mdhip_array *a_in, const mdhip_array *b_in, const mdhip_array *out, int cdt) {
  MD_TRY(md_check_any_array(out, "binary out"));
  if (!a_in->is_scalar) MD_TRY(md_check_any_array(a_in, "binary a"));
  if (!b_in->is_scalar) MD_TRY(md_check_any_array(b_in, "binary b"));
  if (cdt < 0 || cdt >= MDHIP_NUM_ALL_DTYPES) return md_fail(MDHIP_ETYPE, "binary: bad compute dtype %d", cdt);
  const mdhip_array av = md_scalar_in_loop_dtype(a_in, cdt), bv = md_scalar_in_loop_dtype(b_in, cdt);
  const mdhip_array *a = &av, *b = &bv;
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
  if (cdt == MDHIP_BOOL && !is_log && !is_cmp) return md_fail(MDHIP_ETYPE, "bool loops take the wide entry point");
  // uint64 against a SIGNED operand (NumPy's 'Qq->?' / 'qQ->?' comparison loops; a Python int against a uint64 array): compared
  // mathematically — in 128 bits, where both ranges fit — instead of after a cast of one side to the other's type
  const bool a_u64 = a->dtype == MDHIP_U64, b_u64 = b->dtype == MDHIP_U64;
  // (a scalar operand counts as signed only when it IS negative: `u64_array < 5` keeps the uint64 stream kernel)
  const bool a_sgn = md_dtype_is_int(a->dtype) && !md_dtype_is_unsigned(a->dtype) && !(a->is_scalar && a->scalar_i >= 0);
  const bool b_sgn = md_dtype_is_int(b->dtype) && !md_dtype_is_unsigned(b->dtype) && !(b->is_scalar && b->scalar_i >= 0);
  if (is_cmp && ((a_u64 && b_sgn) || (b_u64 && a_sgn))) {
    switch (op) {
      case MDHIP_B_EQ: return X::template nbinary<BEq, __int128>(it, a, b, out);
      case MDHIP_B_NE: return X::template nbinary<BNe, __int128>(it, a, b, out);
      case MDHIP_B_LT: return X::template nbinary<BLt, __int128>(it, a, b, out);
      case MDHIP_B_LE: return X::template nbinary<BLe, __int128>(it, a, b, out);
      case MDHIP_B_GT: return X::template nbinary<BGt, __int128>(it, a, b, out);
      default: return X::template nbinary<BGe, __int128>(it, a, b, out);
    }
  }
  switch (op) {
#define MD_NB_NUM(code, F) \
  case code: MD_CARRIER_NUM_SWITCH(cdt, T, return (X::template nbinary<F, T>(it, a, b, out)))
    MD_NB_NUM(MDHIP_B_ADD, BAdd)
    MD_NB_NUM(MDHIP_B_SUB, BSub)
    MD_NB_NUM(MDHIP_B_MUL, BMul)
    MD_NB_NUM(MDHIP_B_FLOOR_DIV, BFloorDiv)
    MD_NB_NUM(MDHIP_B_MOD, BMod)
    MD_NB_NUM(MDHIP_B_POW, BPow)
    MD_NB_NUM(MDHIP_B_MAXIMUM, BMaximum)
    MD_NB_NUM(MDHIP_B_MINIMUM, BMinimum)
#undef MD_NB_NUM
    case MDHIP_B_TRUE_DIV: MD_CARRIER_FLOAT_SWITCH(cdt, T, return (X::template nbinary<BTrueDiv, T>(it, a, b, out)))
#define MD_NB_CMP(code, F) \
  case code: MD_CARRIER_ALL_SWITCH(cdt, T, return (X::template nbinary<F, T>(it, a, b, out)))
    MD_NB_CMP(MDHIP_B_EQ, BEq)
    MD_NB_CMP(MDHIP_B_NE, BNe)
    MD_NB_CMP(MDHIP_B_LT, BLt)
    MD_NB_CMP(MDHIP_B_LE, BLe)
    MD_NB_CMP(MDHIP_B_GT, BGt)
    MD_NB_CMP(MDHIP_B_GE, BGe)
#undef MD_NB_CMP
    case MDHIP_B_LAND: return X::template nbinary<BLand, uint8_t>(it, a, b, out);
    case MDHIP_B_LOR: return X::template nbinary<BLor, uint8_t>(it, a, b, out);
    case MDHIP_B_LXOR: return X::template nbinary<BLxor, uint8_t>(it, a, b, out);
  }
  return md_fail(MDHIP_EVALUE, "unknown binary op code %d", op);
}

// ================================ where ========================================
// out = cond ? a : b with both branches converted to out's dtype (np.result_type of the branches, decided by the caller)
template <class X> int md_narrow_where_dispatch(const mdhip_array *cond, const mdhip_array *a_in, const mdhip_array *b_in, const mdhip_array *out) {
  MD_TRY(md_check_any_array(out, "where out"));
  if (!cond->is_scalar) MD_TRY(md_check_any_array(cond, "where cond"));
  if (!a_in->is_scalar) MD_TRY(md_check_any_array(a_in, "where x"));
  if (!b_in->is_scalar) MD_TRY(md_check_any_array(b_in, "where y"));
  const mdhip_array av = md_scalar_in_loop_dtype(a_in, out->dtype), bv = md_scalar_in_loop_dtype(b_in, out->dtype);
  const mdhip_array *a = &av, *b = &bv;
  MdIter it;
  const mdhip_array *ops[4] = {cond, a, b, out};
  MD_TRY(md_build_iter(&it, 4, ops, out));
  if (it.total == 0) return MDHIP_OK;
  MD_CARRIER_ALL_SWITCH(out->dtype, T, return (X::template nwhere<T>(it, cond, a, b, out)))
}

// ================================ reductions ====================================
// mdhip_reduce with a storage-only RESULT dtype (max / min of int8 -> int8, sum of float16 -> float16 — accumulated in
// float32 and rounded once, as NumPy's pairwise half sum does): reduce into a temporary of the carrier dtype, convert the
// result (n_out elements: small next to the input, which is still read exactly once). Everything else goes straight through.
extern "C" int mdhip_alloc(size_t, void **);
extern "C" int mdhip_free(void *);
extern "C" int mdhip_convert(const mdhip_array *, const mdhip_array *);
template <class X> int md_reduce_any_out(int op, const mdhip_array *x, const mdhip_array *out, uint32_t mask) {
  if (!x || !out) return md_fail(MDHIP_EVALUE, "reduce: null descriptor");
  const int odt = out->dtype;
  const bool direct = !md_is_narrow(odt) || (odt == MDHIP_U64 && (op == MDHIP_R_SUM || op == MDHIP_R_PROD || x->dtype == MDHIP_U64));
  if (direct) return md_reduce_dispatch<X>(op, x, out, mask);
  int cdt;
  switch (odt) {
    case MDHIP_I8: case MDHIP_I16: case MDHIP_U8: case MDHIP_U16: cdt = MDHIP_I32; break;
    case MDHIP_U32: case MDHIP_U64: cdt = MDHIP_I64; break;
    default: cdt = MDHIP_F32; break;   // float16
  }
  MD_TRY(md_check_any_array(out, "reduce out"));
  mdhip_array tmp = *out;
  int64_t n = 1;
  for (int d = out->ndim - 1; d >= 0; --d) { tmp.strides[d] = n; n *= out->shape[d]; }
  void *buf = nullptr;
  MD_TRY(mdhip_alloc((size_t)(n > 0 ? n : 1) * md_dtype_size(cdt), &buf));
  tmp.data = buf;
  tmp.dtype = cdt;
  int rc = md_reduce_dispatch<X>(op, x, &tmp, mask);
  if (rc == MDHIP_OK) rc = mdhip_convert(&tmp, out);
  mdhip_free(buf);   // (stream-ordered on the device: the next user of the block runs after the conversion)
  return rc;
}
