// rng.hip — device-side random fills (opt-in; md_rng.h has the generator and the stream layout).
// Counterpart of the reference's rand / randn / randint / binomial aliases (backend/numpy.py:131-136) for draws that should
// not cross PCIe; the values are NOT NumPy's for a given seed (nothing on a GPU can be) — the default path keeps drawing on the
// host for that reason. HBM-bound on the output: one 4- or 8-byte store per element, ~10 rounds of integer multiplies per 16 bytes.
#include "md_hip.h"
#include "md_rng.h"

namespace {

// One thread = one Philox block (four 32-bit words) = 4 / W consecutive elements of W words each: the generator runs once per
// 16 bytes of random bits and a thread's stores are contiguous (16 B for float32 uniforms).
template <class T, int KIND, int W>
__global__ void __launch_bounds__(MD_BLOCK) k_random_blocks(T *__restrict__ out, int64_t n, uint64_t seed, uint64_t offset, double a, double b) {
  constexpr int EPB = 4 / W;   // elements per block
  const int64_t nblocks = (n + EPB - 1) / EPB, gs = (int64_t)gridDim.x * blockDim.x;
  const uint64_t thr = KIND == MD_RNG_BINOMIAL ? md_bernoulli_threshold(b) : 0;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nblocks; q += gs) {
    const MdPhilox blk = md_philox4x32(offset + (uint64_t)q, seed);
    T v[EPB];
#pragma unroll
    for (int e = 0; e < EPB; ++e) {
      const uint32_t *w = blk.v + e * W;
      if constexpr (KIND == MD_RNG_UNIFORM) v[e] = md_uniform_from(w, T());
      else if constexpr (KIND == MD_RNG_NORMAL) v[e] = md_normal_from(w, T());
      else if constexpr (KIND == MD_RNG_INTEGERS) v[e] = (T)md_integer_from(w, (int64_t)a, (uint64_t)b);
      else v[e] = (T)((uint64_t)w[0] < thr);   // binomial(1, p)
    }
    const int64_t i0 = q * EPB;
#pragma unroll
    for (int e = 0; e < EPB; ++e)
      if (i0 + e < n) out[i0 + e] = v[e];
  }
}

// binomial with n > 1 trials: n words per element, addressed word by word
template <class T>
__global__ void __launch_bounds__(MD_BLOCK) k_random_binomial(T *__restrict__ out, int64_t n, uint64_t seed, uint64_t offset, int64_t trials, double p) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) out[i] = (T)md_rng_binomial(seed, offset, i, trials, p);
}

template <int KIND, class T, int W> int launch(const mdhip_array *out, int64_t n, uint64_t seed, uint64_t offset, double a, double b) {
  k_random_blocks<T, KIND, W><<<md_grid_for((n * W + 3) / 4), MD_BLOCK, 0, md_stream()>>>((T *)out->data, n, seed, offset, a, b);
  return MD_LAUNCH_CHECK("random_fill");
}
template <class T> int launch_binomial(const mdhip_array *out, int64_t n, uint64_t seed, uint64_t offset, double a, double b) {
  if ((int64_t)a == 1) return launch<MD_RNG_BINOMIAL, T, 1>(out, n, seed, offset, a, b);
  k_random_binomial<T><<<md_grid_for(n), MD_BLOCK, 0, md_stream()>>>((T *)out->data, n, seed, offset, (int64_t)a, b);
  return MD_LAUNCH_CHECK("random_fill(binomial)");
}

}  // namespace

extern "C" int mdhip_random_fill(int kind, uint64_t seed, uint64_t offset, double a, double b, const mdhip_array *out) {
  MD_TRY(md_check_array(out, "random out"));
  int64_t n = 1, expect = 1;
  for (int d = out->ndim - 1; d >= 0; --d) {   // C-contiguous outputs only: the stream position of an element is its flat index
    if (out->shape[d] != 1 && out->strides[d] != expect) return md_fail(MDHIP_EVALUE, "random_fill: out must be C-contiguous");
    expect *= out->shape[d];
    n *= out->shape[d];
  }
  if (n == 0) return MDHIP_OK;
  switch (kind) {
    case MD_RNG_UNIFORM:
      if (out->dtype == MDHIP_F32) return launch<MD_RNG_UNIFORM, float, 1>(out, n, seed, offset, a, b);
      if (out->dtype == MDHIP_F64) return launch<MD_RNG_UNIFORM, double, 2>(out, n, seed, offset, a, b);
      break;
    case MD_RNG_NORMAL:
      if (out->dtype == MDHIP_F32) return launch<MD_RNG_NORMAL, float, 2>(out, n, seed, offset, a, b);
      if (out->dtype == MDHIP_F64) return launch<MD_RNG_NORMAL, double, 4>(out, n, seed, offset, a, b);
      break;
    case MD_RNG_INTEGERS:   // a = low, b = span (high - low), both exact in a double up to 2^53
      if (!(b >= 1.0) || b > 9007199254740992.0) return md_fail(MDHIP_EVALUE, "random_fill: integers need 1 <= high - low <= 2^53");
      if (out->dtype == MDHIP_I64) return launch<MD_RNG_INTEGERS, int64_t, 2>(out, n, seed, offset, a, b);
      if (out->dtype == MDHIP_I32) return launch<MD_RNG_INTEGERS, int32_t, 2>(out, n, seed, offset, a, b);
      break;
    case MD_RNG_BINOMIAL:   // a = n trials, b = p
      if (!(a >= 0.0) || a > (double)MD_RNG_BINOMIAL_MAX_N || !(b >= 0.0 && b <= 1.0))
        return md_fail(MDHIP_EVALUE, "random_fill: binomial needs 0 <= n <= %d and 0 <= p <= 1", MD_RNG_BINOMIAL_MAX_N);
      if (out->dtype == MDHIP_I64) return launch_binomial<int64_t>(out, n, seed, offset, a, b);
      if (out->dtype == MDHIP_I32) return launch_binomial<int32_t>(out, n, seed, offset, a, b);
      break;
    default:
      return md_fail(MDHIP_EVALUE, "random_fill: unknown kind %d", kind);
  }
  return md_fail(MDHIP_ETYPE, "random_fill: kind %d cannot fill %s", kind, md_dtype_name(out->dtype));
}
