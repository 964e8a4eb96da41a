// md_ops.h — scalar semantics of every elementwise / reduction operator.
//
// One definition, two compilations: hipcc builds these functors into the gfx950
// kernels (minidiff_amd/csrc/*.hip); g++ builds them into the CPU test double
// under oracle/host_target/. What they restate is the per-element behaviour of
// the NumPy ufunc each reference backend name aliases
// (reference: minidiff/backend/numpy.py:19-95). NumPy's documented inner-loop
// rules that matter here: Python-style floor_divide/remainder (sign of the
// divisor), integer power by repeated squaring with wrap-around, NaN-propagating
// maximum/minimum, sign(NaN)=NaN, integer division by zero -> 0.
#pragma once
// Also compiled at RUN TIME by hiprtc (fusion_jit.hip embeds this file's text in front
// of the generated fused kernels), where no libc / libstdc++ headers exist: keep it
// free of std:: and of host-only constructs.
#if defined(__HIPCC_RTC__)
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#ifndef INFINITY
#define INFINITY __builtin_inff()
#endif
#ifndef INT64_MIN
#define INT64_MIN (-9223372036854775807LL - 1)
#define INT64_MAX 9223372036854775807LL
#define INT32_MIN (-2147483647 - 1)
#define INT32_MAX 2147483647
#endif
#define MD_HD __device__ __forceinline__
#else
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define MD_HD __host__ __device__ __forceinline__
#else
#define MD_HD inline
#endif
#endif

// minimal type utilities (no <type_traits>: see above)
template <bool C, class A, class B> struct md_cond { using type = A; };
template <class A, class B> struct md_cond<false, A, B> { using type = B; };
template <class A, class B> struct md_same { static constexpr bool value = false; };
template <class A> struct md_same<A, A> { static constexpr bool value = true; };

// numpy.bool_ : one byte holding 0 or 1. A distinct type so that conversions to
// and from it normalise (x != 0) instead of truncating.
struct b8 {
  uint8_t v;
};

template <class T> struct md_is_float { static constexpr bool value = false; };
template <> struct md_is_float<float> { static constexpr bool value = true; };
template <> struct md_is_float<double> { static constexpr bool value = true; };

template <class To, class From> struct md_caster {
  static MD_HD To run(From x) { return (To)x; }
};
template <class From> struct md_caster<b8, From> {
  static MD_HD b8 run(From x) { return b8{(uint8_t)(x != (From)0)}; }
};
template <class To> struct md_caster<To, b8> {
  static MD_HD To run(b8 x) { return (To)x.v; }
};
template <> struct md_caster<b8, b8> {
  static MD_HD b8 run(b8 x) { return x; }
};
// float -> integer casts of out-of-range / NaN values are UB in C; NumPy yields
// INT_MIN on x86. Pin that so host double and device agree.
template <> struct md_caster<int64_t, double> {
  static MD_HD int64_t run(double x) {
    if (!(x >= -9223372036854775808.0 && x < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)x;
  }
};
template <> struct md_caster<int64_t, float> {
  static MD_HD int64_t run(float x) { return md_caster<int64_t, double>::run((double)x); }
};
template <> struct md_caster<int32_t, double> {
  static MD_HD int32_t run(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT32_MIN;
    return (int32_t)x;
  }
};
template <> struct md_caster<int32_t, float> {
  static MD_HD int32_t run(float x) { return md_caster<int32_t, double>::run((double)x); }
};
template <class To, class From> MD_HD To md_cast(From x) { return md_caster<To, From>::run(x); }

// ---- math wrappers so f32 stays f32 (OCML accurate versions on device) -------
#define MD_MATH1(name, ffn, dfn)                   \
  MD_HD float md_##name(float x) { return ffn(x); } \
  MD_HD double md_##name(double x) { return dfn(x); }
MD_MATH1(sin, sinf, sin)
MD_MATH1(cos, cosf, cos)
MD_MATH1(tan, tanf, tan)
MD_MATH1(sinh, sinhf, sinh)
MD_MATH1(cosh, coshf, cosh)
MD_MATH1(tanh, tanhf, tanh)
MD_MATH1(exp, expf, exp)
MD_MATH1(log, logf, log)
MD_MATH1(sqrt, sqrtf, sqrt)
MD_MATH1(ceil, ceilf, ceil)
MD_MATH1(floor, floorf, floor)
MD_MATH1(fabs, fabsf, fabs)
#undef MD_MATH1
// sin and cos of one argument with ONE argument reduction (device: OCML sincos — the same reduction and
// polynomials as sinf / cosf, so each result equals the separate call bit for bit; tests/test_lazy_fusion.py)
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
MD_HD void md_sincos(float x, float *s, float *c) { sincosf(x, s, c); }
MD_HD void md_sincos(double x, double *s, double *c) { sincos(x, s, c); }
#else
MD_HD void md_sincos(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
MD_HD void md_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
#endif
MD_HD float md_fmod(float a, float b) { return fmodf(a, b); }
MD_HD double md_fmod(double a, double b) { return fmod(a, b); }
MD_HD float md_pow(float a, float b) { return powf(a, b); }
MD_HD double md_pow(double a, double b) { return pow(a, b); }
MD_HD float md_copysign(float a, float b) { return copysignf(a, b); }
MD_HD double md_copysign(double a, double b) { return copysign(a, b); }
MD_HD bool md_isnan(float x) { return x != x; }
MD_HD bool md_isnan(double x) { return x != x; }
template <class T> MD_HD bool md_isnan(T) { return false; }

// ============================ unary ===========================================
// apply<T>(x) -> T unless the functor declares a bool result (returns b8).
struct UCopy {
  template <class T> static MD_HD T apply(T x) { return x; }
};
struct UAbs {
  template <class T> static MD_HD T apply(T x) {
    if constexpr (md_is_float<T>::value) {
      return md_fabs(x);
    } else if constexpr (sizeof(T) == 1) {
      return x;
    } else {
      using U = typename md_cond<sizeof(T) == 8, uint64_t, uint32_t>::type;
      return x < 0 ? (T)((U)0 - (U)x) : x;
    }
  }
};
struct UNeg {
  template <class T> static MD_HD T apply(T x) {
    if constexpr (md_is_float<T>::value) {
      return -x;
    } else {
      using U = typename md_cond<sizeof(T) == 8, uint64_t, uint32_t>::type;
      return (T)((U)0 - (U)x);
    }
  }
};
struct USign {
  template <class T> static MD_HD T apply(T x) {
    if constexpr (md_is_float<T>::value) {
      if (x != x) return x;
      return (T)((x > (T)0) - (x < (T)0));
    } else {
      return (T)((x > (T)0) - (x < (T)0));
    }
  }
};
struct UCeil {
  template <class T> static MD_HD T apply(T x) {
    if constexpr (md_is_float<T>::value) return md_ceil(x);
    else return x;
  }
};
struct UFloor {
  template <class T> static MD_HD T apply(T x) {
    if constexpr (md_is_float<T>::value) return md_floor(x);
    else return x;
  }
};
#define MD_UFLOAT(Name, fn)                                        \
  struct Name {                                                    \
    template <class T> static MD_HD T apply(T x) { return fn(x); } \
  };
MD_UFLOAT(USin, md_sin)
MD_UFLOAT(UCos, md_cos)
MD_UFLOAT(UTan, md_tan)
MD_UFLOAT(USinh, md_sinh)
MD_UFLOAT(UCosh, md_cosh)
MD_UFLOAT(UTanh, md_tanh)
MD_UFLOAT(UExp, md_exp)
MD_UFLOAT(ULog, md_log)
MD_UFLOAT(USqrt, md_sqrt)
#undef MD_UFLOAT
// power with a host-scalar exponent is dispatched to these (same values as BPow's shortcuts)
struct USquare {
  template <class T> static MD_HD T apply(T x) { return x * x; }
};
struct URecip {
  template <class T> static MD_HD T apply(T x) { return (T)1 / x; }
};
struct UOne {
  template <class T> static MD_HD T apply(T) { return (T)1; }
};
struct UPowHalf {
  template <class T> static MD_HD T apply(T a) {
    if (a > (T)0 && a != (T)INFINITY) return md_sqrt(a);
    if (a == (T)0) return (T)0;
    return md_pow(a, (T)0.5);
  }
};
struct UInvert {  // integers: bitwise not. (bool is routed to ULogicalNot.)
  template <class T> static MD_HD T apply(T x) { return (T)~x; }
};
// bool-valued unaries
struct ULogicalNot {
  template <class T> static MD_HD b8 apply(T x) { return b8{(uint8_t)(x == (T)0)}; }
};
struct UIsnan {
  template <class T> static MD_HD b8 apply(T x) { return b8{(uint8_t)md_isnan(x)}; }
};

// ============================ binary ==========================================
template <class T> struct md_uint_of {
  using type = typename md_cond<sizeof(T) == 8, uint64_t,
               typename md_cond<sizeof(T) == 4, uint32_t, uint8_t>::type>::type;
};
struct BAdd {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) return a + b;
    else { using U = typename md_uint_of<T>::type; return (T)((U)a + (U)b); }
  }
};
struct BSub {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) return a - b;
    else { using U = typename md_uint_of<T>::type; return (T)((U)a - (U)b); }
  }
};
struct BMul {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) return a * b;
    else { using U = typename md_uint_of<T>::type; return (T)((U)a * (U)b); }
  }
};
struct BTrueDiv {
  template <class T> static MD_HD T apply(T a, T b) { return a / b; }
};
// NumPy npy_divmod (floats) / floor_div_@TYPE@ (ints): result takes the sign of
// the divisor; integer x // 0 == 0 and x % 0 == 0.
template <class T> MD_HD T md_float_divmod(T a, T b, T *modulus) {
  T mod = md_fmod(a, b);
  if (b == (T)0) {
    *modulus = mod;
    return a / b;
  }
  T div = (a - mod) / b;
  if (mod != (T)0) {
    if ((b < (T)0) != (mod < (T)0)) {
      mod += b;
      div -= (T)1;
    }
  } else {
    mod = md_copysign((T)0, b);
  }
  T floordiv;
  if (div != (T)0) {
    floordiv = md_floor(div);
    if (div - floordiv > (T)0.5) floordiv += (T)1;
  } else {
    floordiv = md_copysign((T)0, a / b);
  }
  *modulus = mod;
  return floordiv;
}
struct BFloorDiv {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) {
      T m;
      return md_float_divmod(a, b, &m);
    } else {
      if (b == 0) return 0;
      if (b == (T)-1) { using U = typename md_uint_of<T>::type; return (T)((U)0 - (U)a); }
      T q = a / b;
      if (((a > 0) != (b > 0)) && (q * b != a)) q -= 1;
      return q;
    }
  }
};
struct BMod {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) {
      T m;
      md_float_divmod(a, b, &m);
      return m;
    } else {
      if (b == 0) return 0;
      if (b == (T)-1) return 0;
      T r = a % b;
      if (r != 0 && ((r < 0) != (b < 0))) r += b;
      return r;
    }
  }
};
struct BPow {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) {
      // exact shortcuts for the exponents the tape produces (x**2, x**1, x**0.5)
      if (b == (T)2) return a * a;
      if (b == (T)1) return a;
      if (b == (T)0) return (T)1;
      if (b == (T)0.5) {  // pow(+-0, .5) = +0; sqrt(-0) would be -0
        if (a > (T)0 && a != (T)INFINITY) return md_sqrt(a);
        if (a == (T)0) return (T)0;
      }
      if (b == (T)-1) return (T)1 / a;
      return md_pow(a, b);
    } else {
      // npy integer power: square-and-multiply, wraps; negative exponents are
      // rejected before launch (ValueError in NumPy).
      using U = typename md_uint_of<T>::type;
      if (b == 0) return 1;
      if (a == 1) return 1;
      if (b < 0) return 0;
      U base = (U)a, acc = 1;
      U e = (U)b;
      while (e) {
        if (e & 1) acc *= base;
        base *= base;
        e >>= 1;
      }
      return (T)acc;
    }
  }
};
struct BMaximum {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) return (a >= b || a != a) ? a : b;
    else return a >= b ? a : b;
  }
};
struct BMinimum {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) return (a <= b || a != a) ? a : b;
    else return a <= b ? a : b;
  }
};
#define MD_BCMP(Name, expr)                                                           \
  struct Name {                                                                       \
    template <class T> static MD_HD b8 apply(T a, T b) { return b8{(uint8_t)(expr)}; } \
  };
MD_BCMP(BEq, a == b)
MD_BCMP(BNe, a != b)
MD_BCMP(BLt, a < b)
MD_BCMP(BLe, a <= b)
MD_BCMP(BGt, a > b)
MD_BCMP(BGe, a >= b)
// logical ops run on truth values (compute type uint8 holding 0/1)
MD_BCMP(BLand, (a != (T)0) && (b != (T)0))
MD_BCMP(BLor, (a != (T)0) || (b != (T)0))
MD_BCMP(BLxor, (a != (T)0) != (b != (T)0))
#undef MD_BCMP

// ============================ reductions ======================================
// combine(acc, x) must be associative up to fp rounding; identity<T>() seeds it.
struct RSum {
  template <class T> static MD_HD T identity() { return (T)0; }
  template <class T> static MD_HD T combine(T a, T b) { return BAdd::apply(a, b); }
};
struct RProd {
  template <class T> static MD_HD T identity() { return (T)1; }
  template <class T> static MD_HD T combine(T a, T b) { return BMul::apply(a, b); }
};
template <class T> MD_HD T md_lowest() {
  if constexpr (md_is_float<T>::value) return (T)-INFINITY;
  else if constexpr (sizeof(T) == 8) return (T)INT64_MIN;
  else if constexpr (sizeof(T) == 4) return (T)INT32_MIN;
  else return (T)0;
}
template <class T> MD_HD T md_highest() {
  if constexpr (md_is_float<T>::value) return (T)INFINITY;
  else if constexpr (sizeof(T) == 8) return (T)INT64_MAX;
  else if constexpr (sizeof(T) == 4) return (T)INT32_MAX;
  else return (T)1;
}
// (device, float: IEEE-754 maximum/minimum — one v_maximum3_f32 / v_minimum3_f32 on gfx950 instead of
// two NaN tests and a select; the reductions were VALU-limited on it)
struct RMax {  // NaN-propagating like np.max
  template <class T> static MD_HD T identity() { return md_lowest<T>(); }
  template <class T> static MD_HD T combine(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (md_same<T, float>::value) return __builtin_elementwise_maximum(a, b);
#endif
    if constexpr (md_is_float<T>::value) {
      if (a != a) return a;
      if (b != b) return b;
    }
    return a >= b ? a : b;
  }
};
struct RMin {
  template <class T> static MD_HD T identity() { return md_highest<T>(); }
  template <class T> static MD_HD T combine(T a, T b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (md_same<T, float>::value) return __builtin_elementwise_minimum(a, b);
#endif
    if constexpr (md_is_float<T>::value) {
      if (a != a) return a;
      if (b != b) return b;
    }
    return a <= b ? a : b;
  }
};
struct RAny {  // on truth values
  template <class T> static MD_HD T identity() { return (T)0; }
  template <class T> static MD_HD T combine(T a, T b) { return (T)((a != (T)0) || (b != (T)0)); }
};
struct RAll {
  template <class T> static MD_HD T identity() { return (T)1; }
  template <class T> static MD_HD T combine(T a, T b) { return (T)((a != (T)0) && (b != (T)0)); }
};

// arg-reductions carry (value, position); first occurrence wins, NaN wins outright
template <class T> struct md_argpair {
  T v;
  int64_t i;
};
template <bool IsMax> struct RArg {
  template <class T> static MD_HD md_argpair<T> identity() {
    return md_argpair<T>{IsMax ? md_lowest<T>() : md_highest<T>(), INT64_MAX};
  }
  template <class T> static MD_HD bool better(T a, T b) {  // a strictly better than b; a NaN beats everything but a NaN
    // (one expression, no early returns: in the column walk of the arg-reductions the branchy form became an exec-mask branch
    // per ELEMENT — 442 s_and_saveexec in k_arg_cols_strips<float>)
    if constexpr (md_is_float<T>::value) return (IsMax ? (a > b) : (a < b)) | ((a != a) & (b == b));
    else return IsMax ? (a > b) : (a < b);
  }
  template <class T> static MD_HD md_argpair<T> combine(md_argpair<T> a, md_argpair<T> b) {
    if (b.i == INT64_MAX) return a;
    if (a.i == INT64_MAX) return b;
    if (better(a.v, b.v)) return a;
    if (better(b.v, a.v)) return b;
    return a.i <= b.i ? a : b;
  }
};
