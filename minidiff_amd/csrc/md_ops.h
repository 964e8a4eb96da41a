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
typedef short int16_t;
typedef unsigned short uint16_t;
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

template <class T> struct md_is_unsigned { static constexpr bool value = false; };
template <> struct md_is_unsigned<uint8_t> { static constexpr bool value = true; };
template <> struct md_is_unsigned<uint16_t> { static constexpr bool value = true; };
template <> struct md_is_unsigned<uint32_t> { static constexpr bool value = true; };
template <> struct md_is_unsigned<uint64_t> { static constexpr bool value = true; };

// IEEE binary16 as a STORAGE type (numpy.float16): arithmetic on it runs in float32 and rounds once on the way back, as NumPy's
// own half loops do. Host builds convert by bit manipulation (no _Float16 in the host toolchain), device builds in hardware.
struct f16 {
  uint16_t bits;
};
MD_HD double md_half_to_double(uint16_t h) {
  const uint32_t sign = (h >> 15) & 1u, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
  double v;
  if (e == 0) v = (double)m * 5.9604644775390625e-08;                 // subnormal: m * 2^-24
  else if (e == 31) { v = m ? __builtin_nan("") : __builtin_inf(); }
  else {
    uint64_t bits = ((uint64_t)(e + 1008u) << 52) | ((uint64_t)m << 42);   // exponent bias 1023 - 15
    __builtin_memcpy(&v, &bits, 8);
  }
  return sign ? -v : v;
}
// round-to-nearest-even straight from double (a float widens to double exactly, so one routine serves both without double rounding)
MD_HD uint16_t md_double_to_half(double d) {
  uint64_t b;
  __builtin_memcpy(&b, &d, 8);
  const uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
  const int64_t e = (int64_t)((b >> 52) & 0x7FF) - 1023;
  uint64_t m = b & 0xFFFFFFFFFFFFFull;
  if (e == 1024) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0u));           // inf / nan
  if (e > 15) return (uint16_t)(sign | 0x7C00u);                                    // overflow -> inf
  if (e >= -14) {                                                                   // normal half
    uint64_t q = m >> 42, rem = m & ((1ull << 42) - 1), half = 1ull << 41;
    uint32_t r = (uint32_t)(((uint64_t)(e + 15) << 10) | q);
    if (rem > half || (rem == half && (r & 1u))) ++r;                               // carries into the exponent correctly, up to inf
    return (uint16_t)(sign | r);
  }
  if (e < -25) return sign;                                                         // below half of the smallest subnormal
  m |= 1ull << 52;                                                                  // subnormal half: value = m * 2^(e-52), unit 2^-24
  const int shift = (int)(28 - e);                                                  // 52 - (e + 24)
  uint64_t q = m >> shift, rem = m & ((1ull << shift) - 1), half = 1ull << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  return (uint16_t)(sign | (uint16_t)q);
}
MD_HD float md_f16_to_float(f16 h) {
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
  _Float16 v;
  __builtin_memcpy(&v, &h.bits, 2);
  return (float)v;
#else
  return (float)md_half_to_double(h.bits);
#endif
}
MD_HD f16 md_float_to_f16(float x) {
  f16 h;
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
  const _Float16 v = (_Float16)x;   // round-to-nearest-even (v_cvt_f16_f32)
  __builtin_memcpy(&h.bits, &v, 2);
#else
  h.bits = md_double_to_half((double)x);
#endif
  return h;
}

template <class To, class From> struct md_caster {
  static MD_HD To run(From x) { return (To)x; }
};
template <class To> struct md_caster<To, f16> {
  static MD_HD To run(f16 x) { return (To)md_f16_to_float(x); }
};
template <class From> struct md_caster<f16, From> {
  static MD_HD f16 run(From x) { return md_float_to_f16((float)x); }
};
template <> struct md_caster<f16, double> {
  static MD_HD f16 run(double x) { f16 h; h.bits = md_double_to_half(x); return h; }
};
template <> struct md_caster<f16, f16> {
  static MD_HD f16 run(f16 x) { return x; }
};
template <> struct md_caster<f16, b8> {
  static MD_HD f16 run(b8 x) { return md_float_to_f16((float)x.v); }
};
template <> struct md_caster<b8, f16> {
  static MD_HD b8 run(f16 x) { return b8{(uint8_t)((x.bits & 0x7FFFu) != 0)}; }
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
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
// float32 sin / cos on the device. OCML's sinf / cosf decide small / large argument PER ELEMENT with an exec-masked branch each
// and carry the Payne-Hanek path inline: the streaming `sin` of 1e8 elements issued ~60 vector instructions per element and ran
// VALU-bound at 68-70 % of the HBM roofline (rocprofv3, profiles/r3_cfg3_kernel_stats.csv; ISA: 753 vector instructions per 4
// elements). Here: ONE branch-free path for |x| <= 105615 — three-constant Cody-Waite reduction by pi/2 with fused multiply-adds,
// minimax polynomials on [-pi/4, pi/4], both polynomials evaluated and chosen by quadrant — 21 instructions for sin AND cos
// together, maximum error 1.5 ulp (checked against float64 over 1.7e7 arguments incl. every k pi/2 up to 6e4 within 3e-7
// relative: sincos_check in profiles/r4_sincos.txt; NumPy's own SIMD float32 loops are specified to <= 4 ulp); larger
// arguments, infinities and NaN take OCML's routine. sin, cos and sincos share the routine, so sin(v) and cos(v) of a fused
// backward pass are bit-identical to the eager kernels' (tests/test_lazy_fusion.py).
MD_HD void md_sincos_small(float x, float *sn, float *cs) {
  float j = __builtin_fmaf(x, 0.636619747f, 12582912.0f) - 12582912.0f;   // rint(x * 2 / pi)
  float r = __builtin_fmaf(j, -1.57079601e+00f, x);
  r = __builtin_fmaf(j, -3.13916473e-07f, r);
  r = __builtin_fmaf(j, -5.39030253e-15f, r);
  const int i = (int)j;
  const float s = r * r;
  float p = 2.86567956e-6f;
  p = __builtin_fmaf(p, s, -1.98559923e-4f);
  p = __builtin_fmaf(p, s, 8.33338592e-3f);
  p = __builtin_fmaf(p, s, -1.66666672e-1f);
  const float sp = __builtin_fmaf(p, __builtin_fmaf(r, s, 0.0f), r);      // sin(r); (r * s + 0: a -0 argument comes out as -0)
  float q = 2.44677067e-5f;
  q = __builtin_fmaf(q, s, -1.38877297e-3f);
  q = __builtin_fmaf(q, s, 4.16666567e-2f);
  q = __builtin_fmaf(q, s, -0.5f);
  const float cp = __builtin_fmaf(q, s, 1.0f);       // cos(r)
  float a = (i & 1) ? cp : sp, b = (i & 1) ? sp : cp;
  a = (i & 2) ? -a : a;
  b = ((i + 1) & 2) ? -b : b;
  *sn = a;
  *cs = b;
}
MD_HD void md_sincos(float x, float *s, float *c) {
  if (__builtin_fabsf(x) <= 105615.0f) md_sincos_small(x, s, c);
  else sincosf(x, s, c);   // (also NaN and the infinities: the comparison is false for them)
}
MD_HD void md_sincos(double x, double *s, double *c) { sincos(x, s, c); }
MD_HD float md_sin(float x) { float s, c; md_sincos(x, &s, &c); return s; }
MD_HD float md_cos(float x) { float s, c; md_sincos(x, &s, &c); return c; }
MD_HD double md_sin(double x) { return sin(x); }
MD_HD double md_cos(double x) { return cos(x); }
#else
MD_MATH1(sin, sinf, sin)
MD_MATH1(cos, cosf, cos)
MD_HD void md_sincos(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
MD_HD void md_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
#endif
#undef MD_MATH1
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
               typename md_cond<sizeof(T) == 4, uint32_t,
               typename md_cond<sizeof(T) == 2, uint16_t, uint8_t>::type>::type>::type;
};
struct BAdd {
  template <class T> static MD_HD T apply(T a, T b) {
    if constexpr (md_is_float<T>::value) return a + b;
    else { using U = typename md_uint_of<T>::type; return (T)((U)a + (U)b); }
  }
};
// a += b in the array's own storage type (np.add.at): integers wrap, float16 rounds after EVERY addition
template <class T> MD_HD T md_storage_add(T a, T b) {
  if constexpr (md_same<T, f16>::value) return md_float_to_f16(md_f16_to_float(a) + md_f16_to_float(b));
  else return BAdd::apply(a, b);
}
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
      if constexpr (md_is_unsigned<T>::value) {
        return a / b;
      } else {
        if (b == (T)-1) { using U = typename md_uint_of<T>::type; return (T)((U)0 - (U)a); }
        T q = a / b;
        if (((a > 0) != (b > 0)) && (q * b != a)) q -= 1;
        return q;
      }
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
      if constexpr (md_is_unsigned<T>::value) {
        return a % b;
      } else {
        if (b == (T)-1) return 0;
        T r = a % b;
        if (r != 0 && ((r < 0) != (b < 0))) r += b;
        return r;
      }
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
      if constexpr (!md_is_unsigned<T>::value) {
        if (b < 0) return 0;
      }
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
  else if constexpr (md_is_unsigned<T>::value && sizeof(T) == 8) return (T)0;
  else if constexpr (sizeof(T) == 8) return (T)INT64_MIN;
  else if constexpr (sizeof(T) == 4) return (T)INT32_MIN;
  else return (T)0;
}
template <class T> MD_HD T md_highest() {
  if constexpr (md_is_float<T>::value) return (T)INFINITY;
  else if constexpr (md_is_unsigned<T>::value && sizeof(T) == 8) return (T)~(uint64_t)0;
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
