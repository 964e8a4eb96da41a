// md_rng.h — counter-based random numbers (Philox4x32-10, Salmon et al., SC'11) shared by the device kernels and the CPU test
// double, so that both produce the SAME stream for a (seed, offset) pair. Opt-in counterpart of the reference's np.random.* aliases
// (backend/numpy.py:129-137): the default path still draws on the host with NumPy, because a given np.random.seed must give the
// reference's numbers; this one (MDHIP_DEVICE_RNG=1 / ndarray.device_rng) keeps large draws off PCIe.
// Stream layout: draw number i of a call comes from the Philox block with counter (offset + i / 4), word i % 4 (two consecutive words
// for a 53-bit double or a 64-bit key: draw i uses block offset + i / 2, words 2 * (i % 2) and + 1); the caller advances `offset`
// by the number of blocks a call may touch (md_rng_blocks).
#pragma once
#include <stdint.h>

#include "md_ops.h"

struct MdPhilox { uint32_t v[4]; };

MD_HD uint32_t md_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
MD_HD uint64_t md_mulhi64(uint64_t a, uint64_t b) {
  const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
  const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
  const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
  return p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
}

MD_HD MdPhilox md_philox4x32(uint64_t counter, uint64_t seed) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = 0x6d646870u /* "mdhp" */, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = md_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = md_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  MdPhilox o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// uniform in [0, 1): 24 bits for float, 53 for double (exact conversions: device and host agree bit for bit)
MD_HD float md_u01f(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
MD_HD double md_u01d(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }

// kinds of mdhip_random_fill
enum { MD_RNG_UNIFORM = 0, MD_RNG_NORMAL = 1, MD_RNG_INTEGERS = 2, MD_RNG_BINOMIAL = 3 };
constexpr int MD_RNG_BINOMIAL_MAX_N = 256;   // exact sum of Bernoulli draws up to here

// word j of the stream that starts at block `offset`
MD_HD uint32_t md_rng_word(uint64_t seed, uint64_t offset, uint64_t j) { return md_philox4x32(offset + (j >> 2), seed).v[j & 3]; }

// ---- values from the words of an element (the device kernels compute a Philox block ONCE and feed its words here) ---------
template <class T> struct MdRngWords;   // words per element of the float kinds
template <> struct MdRngWords<float> { static constexpr int uniform = 1, normal = 2; };
template <> struct MdRngWords<double> { static constexpr int uniform = 2, normal = 4; };
MD_HD float md_uniform_from(const uint32_t *w, float) { return md_u01f(w[0]); }
MD_HD double md_uniform_from(const uint32_t *w, double) { return md_u01d(w[0], w[1]); }
// standard normal by Box-Muller in the output precision (log / cos of the platform's libm: device and host agree to a few ulp;
// the second Box-Muller output is not kept: a fixed number of words per element keeps the stream indexable)
MD_HD float md_normal_from(const uint32_t *w, float) {
  const float u1 = 1.0f - md_u01f(w[0]), u2 = md_u01f(w[1]);   // u1 in (0, 1]
  return md_sqrt(-2.0f * md_log(u1)) * md_cos(6.283185307179586f * u2);
}
MD_HD double md_normal_from(const uint32_t *w, double) {
  const double u1 = 1.0 - md_u01d(w[0], w[1]), u2 = md_u01d(w[2], w[3]);
  return md_sqrt(-2.0 * md_log(u1)) * md_cos(6.283185307179586 * u2);
}
// integer in [lo, lo + span): 64 random bits, multiply-shift (bias < span / 2^64)
MD_HD int64_t md_integer_from(const uint32_t *w, int64_t lo, uint64_t span) { return lo + (int64_t)md_mulhi64(((uint64_t)w[0] << 32) | w[1], span); }
// p as a 32-bit threshold for Bernoulli trials (p = 1 -> every trial succeeds)
MD_HD uint64_t md_bernoulli_threshold(double p) { return p >= 1.0 ? 0x100000000ull : p <= 0.0 ? 0ull : (uint64_t)(p * 4294967296.0); }

// ---- the same, addressed by element index (CPU test double; device paths without a whole-block layout) -----------------------
template <class T> MD_HD T md_rng_uniform(uint64_t seed, uint64_t offset, int64_t i) {
  constexpr int W = MdRngWords<T>::uniform;
  uint32_t w[W];
  for (int k = 0; k < W; ++k) w[k] = md_rng_word(seed, offset, (uint64_t)W * (uint64_t)i + k);
  return md_uniform_from(w, T());
}
template <class T> MD_HD T md_rng_normal(uint64_t seed, uint64_t offset, int64_t i) {
  constexpr int W = MdRngWords<T>::normal;
  uint32_t w[W];
  for (int k = 0; k < W; ++k) w[k] = md_rng_word(seed, offset, (uint64_t)W * (uint64_t)i + k);
  return md_normal_from(w, T());
}
MD_HD int64_t md_rng_integer(uint64_t seed, uint64_t offset, int64_t i, int64_t lo, uint64_t span) {
  const uint32_t w[2] = {md_rng_word(seed, offset, 2 * (uint64_t)i), md_rng_word(seed, offset, 2 * (uint64_t)i + 1)};
  return md_integer_from(w, lo, span);
}
// binomial(n, p), n <= MD_RNG_BINOMIAL_MAX_N: the number of the element's n words below the threshold
MD_HD int64_t md_rng_binomial(uint64_t seed, uint64_t offset, int64_t i, int64_t n, double p) {
  int64_t c = 0;
  const uint64_t thr = md_bernoulli_threshold(p);
  for (int64_t t = 0; t < n; ++t) c += (uint64_t)md_rng_word(seed, offset, (uint64_t)i * (uint64_t)n + (uint64_t)t) < thr;
  return c;
}
// 64-bit sort key of element i (permutation: sort indices by key)
MD_HD uint64_t md_rng_key(uint64_t seed, uint64_t offset, int64_t i) {
  return ((uint64_t)md_rng_word(seed, offset, 2 * (uint64_t)i) << 32) | md_rng_word(seed, offset, 2 * (uint64_t)i + 1);
}
