// moments.hip — variance / standard deviation along one axis in ONE entry point (mdhip_var).
//
// Serves reference minidiff/backend/numpy.py:57 (np.std) as called by minidiff/ops/definitions.py:209-221 (forward AND the vjp,
// which evaluates std(x) again). NumPy's _methods._var is mean -> x - mean -> square -> sum -> divide (-> sqrt): composed from the
// other entry points that is four reads and two writes of the array (977 GB/s of algorithmic bytes on 8192 x 4096 float32). Here:
//   rows    (the reduced axis is the contiguous one): a row of up to 16 Ki elements is read ONCE into registers — sum, mean (a true
//           division, as NumPy), centred squares, all from the registers; longer rows are read twice (the second time from cache);
//   columns (axis 0 of a row-major matrix): the column sums by mdhip_reduce's strips kernel, then one more strips walk that adds
//           (x - mean)^2 — two reads, the last block of a strip folds the band partials in band order and finishes the division.
// Same arithmetic as NumPy up to the order of the two sums (no Welford update, no E[x^2] - mean^2 cancellation). HBM-bound:
// algorithmic bytes = one read of x.
#include "md_hip.h"

extern "C" int mdhip_alloc(size_t, void **);
extern "C" int mdhip_free(void *);
extern "C" int mdhip_reduce(int op, const mdhip_array *x, const mdhip_array *out, uint32_t axis_mask);

namespace {

template <class T> __device__ __forceinline__ T md_shfl_xor_t(T v, int m) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int w[2]; } u;
    u.t = v;
    u.w[0] = __shfl_xor(u.w[0], m, 64);
    u.w[1] = __shfl_xor(u.w[1], m, 64);
    return u.t;
  } else {
    union { T t; int w; } u;
    u.t = v;
    u.w = __shfl_xor(u.w, m, 64);
    return u.t;
  }
}
template <class T> __device__ __forceinline__ T wave_sum_all(T v) {   // every lane gets the sum (fixed butterfly)
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += md_shfl_xor_t(v, m);
  return v;
}

// One row per GROUP of G lanes-worth of threads: G = 64 (a wave, four rows per block) or 256 (the block). NV: 16-B vectors of the
// row cached per thread (0: the row does not fit, read it twice).
template <class T, int G, int NV>
__global__ void __launch_bounds__(256) k_var_rows(const T *__restrict__ x, int64_t n_rows, int64_t n, T *__restrict__ out, T denom, int take_sqrt) {
  constexpr int V = 16 / sizeof(T);
  typedef MdVec<T, V> Vec;
  __shared__ T red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = G == 64 ? (int64_t)blockIdx.x * 4 + w : (int64_t)blockIdx.x;
  const int tig = G == 64 ? lane : (int)threadIdx.x;   // thread in group
  const bool live = row < n_rows;
  const Vec *p = reinterpret_cast<const Vec *>(x + (live ? row : 0) * n);
  const int64_t nvec = n / V;
  auto group_sum = [&](T v) -> T {
    v = wave_sum_all(v);
    if constexpr (G == 256) {
      __syncthreads();   // (red is reused)
      if (lane == 0) red[w] = v;
      __syncthreads();
      v = (red[0] + red[1]) + (red[2] + red[3]);
    }
    return v;
  };
  T s = (T)0;
  if constexpr (NV > 0) {
    Vec c[NV];
#pragma unroll
    for (int g = 0; g < NV; ++g) {
      const int64_t i = tig + (int64_t)g * G;
      if (i < nvec) c[g] = p[i];
      else {
#pragma unroll
        for (int j = 0; j < V; ++j) c[g].v[j] = (T)0;
      }
    }
#pragma unroll
    for (int g = 0; g < NV; ++g)
#pragma unroll
      for (int j = 0; j < V; ++j) s += c[g].v[j];
    const T mean = group_sum(s) / (T)n;
    T q = (T)0;
#pragma unroll
    for (int g = 0; g < NV; ++g) {
      const int64_t i = tig + (int64_t)g * G;
      if (i < nvec) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const T d = c[g].v[j] - mean;
          q = fma(d, d, q);
        }
      }
    }
    q = group_sum(q);
    if (live && tig == 0) {
      const T v = q / denom;
      out[row] = take_sqrt ? sqrt(v) : v;
    }
  } else {
    for (int64_t i = tig; i < nvec; i += G) {
      const Vec t = p[i];
#pragma unroll
      for (int j = 0; j < V; ++j) s += t.v[j];
    }
    const T mean = group_sum(s) / (T)n;
    T q = (T)0;
    for (int64_t i = tig; i < nvec; i += G) {
      const Vec t = p[i];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const T d = t.v[j] - mean;
        q = fma(d, d, q);
      }
    }
    q = group_sum(q);
    if (live && tig == 0) {
      const T v = q / denom;
      out[row] = take_sqrt ? sqrt(v) : v;
    }
  }
}

// Second pass of the column form: out[c] = f( sum_r (x[r][c] - csum[c] / n)^2 / denom ). The walk of k_reduce_cols_strips (reduce.hip):
// NS strips x NB bands, rows interleaved across bands and waves, batches of RB rows double-buffered, ticket finish.
template <class T, int RB>
__global__ void __launch_bounds__(256) k_var_cols(const T *__restrict__ x, const T *__restrict__ csum, int64_t n_out, int64_t n_red, int NS, int NB,
                                                  T *partial, unsigned *tickets, T *__restrict__ out, T denom, int take_sqrt) {
  constexpr int V = 16 / sizeof(T);
  typedef MdVec<T, V> Vec;
  __shared__ Vec sm[3][64];
  __shared__ unsigned last_flag;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int s = blockIdx.x % NS, b = blockIdx.x / NS;
  const int64_t col_raw = ((int64_t)s * 64 + lane) * V;
  const bool col_ok = col_raw < n_out;
  const int64_t col = col_ok ? col_raw : n_out - V;
  const int64_t first = b + (int64_t)NB * w, step = (int64_t)NB * 4;
  const int64_t nrw = first < n_red ? (n_red - first + step - 1) / step : 0;
  const int64_t nb = nrw / RB;
  const Vec cs = *reinterpret_cast<const Vec *>(csum + col);
  T mean[V], acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { mean[j] = cs.v[j] / (T)n_red; acc[j] = (T)0; }
  const T *p = x + col + first * n_out;
  const int64_t rstep = step * n_out;
  Vec t[2][RB];
  auto load = [&](int buf, int64_t bt) {
    const int64_t i0 = (bt < nb ? bt : nb - 1) * RB;
#pragma unroll
    for (int u = 0; u < RB; ++u) t[buf][u] = *reinterpret_cast<const Vec *>(p + (i0 + u) * rstep);
  };
  auto add = [&](int buf) {
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const T d = t[buf][u].v[j] - mean[j];
        acc[j] = fma(d, d, acc[j]);
      }
  };
  if (nb > 0) {
    load(0, 0);
    int64_t bt = 0;
    for (; bt + 1 < nb; bt += 2) {
      load(1, bt + 1);
      __builtin_amdgcn_sched_barrier(0);
      add(0);
      __builtin_amdgcn_sched_barrier(0);
      load(0, bt + 2);
      __builtin_amdgcn_sched_barrier(0);
      add(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (bt < nb) add(0);
  }
  for (int64_t i = nb * RB; i < nrw; ++i) {
    const Vec tt = *reinterpret_cast<const Vec *>(p + i * rstep);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const T d = tt.v[j] - mean[j];
      acc[j] = fma(d, d, acc[j]);
    }
  }
  if (w > 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) sm[w - 1][lane].v[j] = acc[j];
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += sm[k][lane].v[j];
  }
  auto finish = [&]() {
    Vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const T v = acc[j] / denom;
      o.v[j] = take_sqrt ? sqrt(v) : v;
    }
    *reinterpret_cast<Vec *>(out + col) = o;
  };
  if (NB == 1) {
    if (w == 0 && col_ok) finish();
    return;
  }
  const __amdgpu_buffer_rsrc_t pr = md_rsrc(partial, (unsigned)((int64_t)NB * n_out * (int64_t)sizeof(T)));
  if (w == 0 && col_ok) {
    Vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = acc[j];
    md_st16_sc1(pr, (unsigned)(((int64_t)b * n_out + col) * (int64_t)sizeof(T)), o);
  }
  if (!md_ticket_last(tickets + s * MD_TICKET_PAD, (unsigned)NB, &last_flag)) return;
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = (T)0;
  for (int r = w; r < NB; r += 4) {
    const Vec pt = md_ld16_sc1<Vec>(pr, (unsigned)(((int64_t)r * n_out + col) * (int64_t)sizeof(T)));
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] += pt.v[j];
  }
  __syncthreads();
  if (w > 0) {
#pragma unroll
    for (int j = 0; j < V; ++j) sm[w - 1][lane].v[j] = acc[j];
  }
  __syncthreads();
  if (w == 0 && col_ok) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] += sm[k][lane].v[j];
    finish();
  }
}

template <class T> int var_rows(const T *x, int64_t n_rows, int64_t n, T *out, T denom, int take_sqrt) {
  constexpr int V = 16 / sizeof(T);
  const int64_t nvec = n / V;
  if (n_rows > (1ll << 31) - 8) return md_fail(MDHIP_EVALUE, "var: too many rows for one launch");
  if (nvec <= 64 * 8) {
    const unsigned grid = (unsigned)((n_rows + 3) / 4);
    if (nvec <= 64 * 2) MD_LAUNCH((k_var_rows<T, 64, 2>), grid, 256, x, n_rows, n, out, denom, take_sqrt);
    else if (nvec <= 64 * 4) MD_LAUNCH((k_var_rows<T, 64, 4>), grid, 256, x, n_rows, n, out, denom, take_sqrt);
    else MD_LAUNCH((k_var_rows<T, 64, 8>), grid, 256, x, n_rows, n, out, denom, take_sqrt);
  } else if (nvec <= 256 * 4) {
    MD_LAUNCH((k_var_rows<T, 256, 4>), (unsigned)n_rows, 256, x, n_rows, n, out, denom, take_sqrt);
  } else if (nvec <= 256 * 16) {
    MD_LAUNCH((k_var_rows<T, 256, 16>), (unsigned)n_rows, 256, x, n_rows, n, out, denom, take_sqrt);
  } else {
    MD_LAUNCH((k_var_rows<T, 256, 0>), (unsigned)n_rows, 256, x, n_rows, n, out, denom, take_sqrt);
  }
  return MD_LAUNCH_CHECK("var(rows)");
}

template <class T> int var_cols(const mdhip_array *x2d, int64_t n_red, int64_t n_out, T *out, T denom, int take_sqrt, int dtype) {
  constexpr int V = 16 / sizeof(T);
  void *csum = nullptr;
  MD_TRY(mdhip_alloc((size_t)n_out * sizeof(T), &csum));
  mdhip_array sd{};
  sd.data = csum; sd.dtype = dtype; sd.ndim = 2; sd.shape[0] = 1; sd.shape[1] = n_out; sd.strides[0] = n_out; sd.strides[1] = 1;
  int rc = mdhip_reduce(MDHIP_R_SUM, x2d, &sd, 1u);
  if (rc != MDHIP_OK) { mdhip_free(csum); return rc; }
  const int64_t NS = (n_out + 64 * V - 1) / (64 * V);
  int64_t NB = (1024 + NS - 1) / NS;
  if (NB > 64) NB = 64;
  if (NB > n_red / 32) NB = n_red / 32;
  if (NB < 1) NB = 1;
  if (NS * MD_TICKET_PAD > MD_TICKET_WORDS) NB = 1;
  while (NB > 1 && NB * n_out * (int64_t)sizeof(T) >= (1ll << 31)) NB /= 2;
  void *partial = nullptr;
  if (NB > 1) {
    rc = mdhip_alloc((size_t)(NB * n_out) * sizeof(T), &partial);
    if (rc != MDHIP_OK) { mdhip_free(csum); return rc; }
  }
  MD_LAUNCH((k_var_cols<T, 8>), (unsigned)(NS * NB), 256, (const T *)x2d->data, (const T *)csum, n_out, n_red, (int)NS, (int)NB, (T *)partial, md_tickets(), out, denom, take_sqrt);
  rc = MD_LAUNCH_CHECK("var(cols)");
  if (partial) mdhip_free(partial);
  mdhip_free(csum);   // stream-ordered
  return rc;
}

}  // namespace

extern "C" int mdhip_var(const mdhip_array *x, const mdhip_array *out, int32_t axis, int64_t ddof, int take_sqrt) {
  MD_TRY(md_check_array(x, "var x"));
  MD_TRY(md_check_array(out, "var out"));
  if (x->is_scalar || out->is_scalar) return md_fail(MDHIP_EVALUE, "var: arrays expected");
  if (x->dtype != MDHIP_F32 && x->dtype != MDHIP_F64) return md_fail(MDHIP_EVALUE, "var: float32 / float64 only (the caller composes the rest)");
  if (out->dtype != x->dtype) return md_fail(MDHIP_ETYPE, "var: out dtype must equal x dtype");
  if (axis < 0 || axis >= x->ndim) return md_fail(MDHIP_EVALUE, "var: axis out of range");
  // x must be C-contiguous: viewed as (outer, n, inner)
  int64_t acc = 1, outer = 1, inner = 1;
  for (int d = x->ndim - 1; d >= 0; --d) {
    if (x->shape[d] != 1 && x->strides[d] != acc) return md_fail(MDHIP_EVALUE, "var: x is not C-contiguous");
    acc *= x->shape[d];
    if (d > axis) inner *= x->shape[d];
    if (d < axis) outer *= x->shape[d];
  }
  const int64_t n = x->shape[axis];
  if (n - ddof <= 0 || n < 2 || outer * inner == 0) return md_fail(MDHIP_EVALUE, "var: degenerate count (the caller composes NumPy's nan / inf)");
  int64_t osz = 1, oacc = 1;
  for (int d = out->ndim - 1; d >= 0; --d) {
    if (out->shape[d] != 1 && out->strides[d] != oacc) return md_fail(MDHIP_EVALUE, "var: out is not C-contiguous");
    oacc *= out->shape[d];
    osz *= out->shape[d];
  }
  if (osz != outer * inner) return md_fail(MDHIP_EVALUE, "var: out has %lld elements, expected %lld", (long long)osz, (long long)(outer * inner));
  const int64_t V = x->dtype == MDHIP_F32 ? 4 : 2;
  if (((uintptr_t)x->data & 15) || ((uintptr_t)out->data & 15)) return md_fail(MDHIP_EVALUE, "var: unaligned operands");
  const double denom = (double)(n - ddof);
  if (inner == 1) {
    if (n % V) return md_fail(MDHIP_EVALUE, "var: row length not a multiple of the 16-B vector");
    // a row is one wave's / one block's work: a few very long rows would leave the chip idle (the composed passes use all of it)
    if (outer < 256 && n > 16384) return md_fail(MDHIP_EVALUE, "var: few long rows (the caller composes)");
    return x->dtype == MDHIP_F32 ? var_rows<float>((const float *)x->data, outer, n, (float *)out->data, (float)denom, take_sqrt)
                                 : var_rows<double>((const double *)x->data, outer, n, (double *)out->data, denom, take_sqrt);
  }
  if (outer == 1) {
    if ((inner % V) || n < 64 || inner < 256) return md_fail(MDHIP_EVALUE, "var: column form needs >= 256 aligned columns and >= 64 rows");
    mdhip_array x2{};
    x2.data = x->data; x2.dtype = x->dtype; x2.ndim = 2; x2.shape[0] = n; x2.shape[1] = inner; x2.strides[0] = inner; x2.strides[1] = 1;
    return x->dtype == MDHIP_F32 ? var_cols<float>(&x2, n, inner, (float *)out->data, (float)denom, take_sqrt, MDHIP_F32)
                                 : var_cols<double>(&x2, n, inner, (double *)out->data, denom, take_sqrt, MDHIP_F64);
  }
  return md_fail(MDHIP_EVALUE, "var: reduced axis in the middle (the caller composes)");
}
