// mdhip_host.cpp — CPU TEST DOUBLE of the libmdhip C-ABI (include/mdhip.h).
//
// TEST INFRASTRUCTURE ONLY. This file is not part of the product: the product
// library is minidiff_amd/libmdhip.so (HIP, gfx950) and the package refuses to
// load anything whose mdhip_target() is not "hip:gfx950" unless a test passes
// the path explicitly. Only tests/ (and the developer, in this GPU-less
// container) load this double, for two purposes:
//   1. exercise the Python shim (views, NumPy promotion, descriptor building,
//      index plans) against NumPy without a GPU, and
//   2. plug the shim into the REAL reference (/root/reference, --backend flag,
//      minidiff/backend/__init__.py:13-19,43-77) to prove the drop-in boundary.
// It shares csrc/md_ops.h + md_dispatch.h with the device build, so dispatch
// and per-element semantics are the same code; the loops below are the naive
// sequential restatement (no tiling, no vectorisation, k-ordered accumulation).
#include <cmath>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <vector>

#include "../../minidiff_amd/csrc/md_dispatch.h"
#include "../../minidiff_amd/csrc/md_narrow.h"
#include "../../minidiff_amd/csrc/md_vm.h"
#include "../../minidiff_amd/csrc/md_rng.h"

std::string &md_err_slot() {
  static thread_local std::string s;
  return s;
}

namespace {
std::mutex g_mu;
std::map<void *, size_t> g_live;
int64_t g_in_use = 0, g_peak = 0, g_nalloc = 0;

struct HostExec {
  template <class F, class Tc, class To>
  static int unary(const MdIter &it, const mdhip_array *x, const mdhip_array *out) {
    Tc sx = x->is_scalar ? md_scalar_as<Tc>(x) : Tc();
    To *o = (To *)out->data;
    int64_t offs[MD_MAX_OPS];
    for (int64_t i = 0; i < it.total; ++i) {
      md_iter_offsets(it, i, offs);
      Tc v = x->is_scalar ? sx : md_load<Tc>(x->data, x->dtype, offs[0]);
      o[offs[1]] = md_to_out<To>(F::apply(v));
    }
    return MDHIP_OK;
  }
  template <class F, class Tc, class To>
  static int binary(const MdIter &it, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
    Tc sa = a->is_scalar ? md_scalar_as<Tc>(a) : Tc();
    Tc sb = b->is_scalar ? md_scalar_as<Tc>(b) : Tc();
    To *o = (To *)out->data;
    int64_t offs[MD_MAX_OPS];
    for (int64_t i = 0; i < it.total; ++i) {
      md_iter_offsets(it, i, offs);
      Tc va = a->is_scalar ? sa : md_load<Tc>(a->data, a->dtype, offs[0]);
      Tc vb = b->is_scalar ? sb : md_load<Tc>(b->data, b->dtype, offs[1]);
      o[offs[2]] = md_to_out<To>(F::apply(va, vb));
    }
    return MDHIP_OK;
  }
  template <class T>
  static int where(const MdIter &it, const mdhip_array *c, const mdhip_array *a, const mdhip_array *b,
                   const mdhip_array *out) {
    uint8_t sc = c->is_scalar ? md_scalar_as<uint8_t>(c) : 0;
    T sa = a->is_scalar ? md_scalar_as<T>(a) : T();
    T sb = b->is_scalar ? md_scalar_as<T>(b) : T();
    T *o = (T *)out->data;
    int64_t offs[MD_MAX_OPS];
    for (int64_t i = 0; i < it.total; ++i) {
      md_iter_offsets(it, i, offs);
      uint8_t vc = c->is_scalar ? sc : md_load<uint8_t>(c->data, c->dtype, offs[0]);
      T va = a->is_scalar ? sa : md_load<T>(a->data, a->dtype, offs[1]);
      T vb = b->is_scalar ? sb : md_load<T>(b->data, b->dtype, offs[2]);
      o[offs[3]] = vc ? va : vb;
    }
    return MDHIP_OK;
  }
  // storage-only dtypes (md_narrow.h): the same generic loops with 12-dtype loads / stores
  template <class F, class Tc> static int nunary(const MdIter &it, const mdhip_array *x, const mdhip_array *out) {
    const Tc sx = x->is_scalar ? md_scalar_as<Tc>(x) : Tc();
    int64_t offs[MD_MAX_OPS];
    for (int64_t i = 0; i < it.total; ++i) {
      md_iter_offsets(it, i, offs);
      const Tc v = x->is_scalar ? sx : md_load<Tc>(x->data, x->dtype, offs[0]);
      md_store_as(out->data, out->dtype, offs[1], F::apply(v));
    }
    return MDHIP_OK;
  }
  template <class F, class Tc> static int nbinary(const MdIter &it, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
    const Tc sa = a->is_scalar ? md_scalar_as<Tc>(a) : Tc(), sb = b->is_scalar ? md_scalar_as<Tc>(b) : Tc();
    int64_t offs[MD_MAX_OPS];
    for (int64_t i = 0; i < it.total; ++i) {
      md_iter_offsets(it, i, offs);
      const Tc va = a->is_scalar ? sa : md_load<Tc>(a->data, a->dtype, offs[0]);
      const Tc vb = b->is_scalar ? sb : md_load<Tc>(b->data, b->dtype, offs[1]);
      md_store_as(out->data, out->dtype, offs[2], F::apply(va, vb));
    }
    return MDHIP_OK;
  }
  template <class Tc> static int nwhere(const MdIter &it, const mdhip_array *c, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
    const uint8_t sc = c->is_scalar ? md_scalar_as<uint8_t>(c) : 0;
    const Tc sa = a->is_scalar ? md_scalar_as<Tc>(a) : Tc(), sb = b->is_scalar ? md_scalar_as<Tc>(b) : Tc();
    int64_t offs[MD_MAX_OPS];
    for (int64_t i = 0; i < it.total; ++i) {
      md_iter_offsets(it, i, offs);
      const uint8_t vc = c->is_scalar ? sc : md_load<uint8_t>(c->data, c->dtype, offs[0]);
      const Tc va = a->is_scalar ? sa : md_load<Tc>(a->data, a->dtype, offs[1]);
      const Tc vb = b->is_scalar ? sb : md_load<Tc>(b->data, b->dtype, offs[2]);
      md_store_as(out->data, out->dtype, offs[3], vc ? va : vb);
    }
    return MDHIP_OK;
  }
  template <class R, class Tacc, class To>
  static int reduce(const MdRedPlan &pl, const mdhip_array *x, const mdhip_array *out) {
    To *o = (To *)out->data;
    for (int64_t i = 0; i < pl.n_out; ++i) {
      int64_t xo, oo;
      md_red_kept_offsets(pl, i, &xo, &oo);
      Tacc acc = R::template identity<Tacc>();
      for (int64_t r = 0; r < pl.n_red; ++r)
        acc = R::combine(acc, md_load<Tacc>(x->data, x->dtype, xo + md_red_offset(pl, r)));
      o[oo] = md_cast<To>(acc);
    }
    return MDHIP_OK;
  }
  template <bool IsMax, class T>
  static int argreduce(const MdRedPlan &pl, const mdhip_array *x, const mdhip_array *out) {
    int64_t *o = (int64_t *)out->data;
    for (int64_t i = 0; i < pl.n_out; ++i) {
      int64_t xo, oo;
      md_red_kept_offsets(pl, i, &xo, &oo);
      md_argpair<T> acc = RArg<IsMax>::template identity<T>();
      for (int64_t r = 0; r < pl.n_red; ++r) {
        md_argpair<T> cur{md_load<T>(x->data, x->dtype, xo + md_red_offset(pl, r)), r};
        acc = RArg<IsMax>::combine(acc, cur);
      }
      o[oo] = acc.i;
    }
    return MDHIP_OK;
  }
  template <class T> static int gemm(const MdGemm &g) {
    const T *A = (const T *)g.a, *B = (const T *)g.b;
    T *C = (T *)g.c;
    for (int64_t bi = 0; bi < g.batch; ++bi)
      for (int64_t m = 0; m < g.M; ++m)
        for (int64_t n = 0; n < g.N; ++n) {
          T acc = (T)0;
          for (int64_t k = 0; k < g.K; ++k) {
            T av = A[bi * g.a_bs + m * g.a_ms + k * g.a_ks];
            T bv = B[bi * g.b_bs + k * g.b_ks + n * g.b_ns];
            if constexpr (md_is_float<T>::value) acc = std::fma(av, bv, acc);
            else acc = BAdd::apply(acc, BMul::apply(av, bv));
          }
          C[bi * g.c_bs + m * g.c_ms + n * g.c_ns] = acc;
        }
    return MDHIP_OK;
  }
};
}  // namespace

extern "C" {

int mdhip_init(int) { return MDHIP_OK; }
int mdhip_device(int *d) { *d = -1; return MDHIP_OK; }
int mdhip_shutdown(void) { return MDHIP_OK; }
// the same option table as the product (csrc/md_options.h): the double's host logic reads it where the product's does
int mdhip_debug_set_option(const char *name, int64_t value) {
  const int id = name ? md_opt_find(name) : -1;
  if (id < 0) return md_fail(MDHIP_EVALUE, "unknown option %s", name ? name : "(null)");
  md_opt_table()[id] = value;
  return MDHIP_OK;
}
int mdhip_debug_get_option(const char *name, int64_t *value_out) {
  const int id = name ? md_opt_find(name) : -1;
  if (id < 0 || !value_out) return md_fail(MDHIP_EVALUE, "unknown option %s", name ? name : "(null)");
  *value_out = md_opt_table()[id];
  return MDHIP_OK;
}
const char *mdhip_target(void) { return "host"; }
const char *mdhip_last_error(void) { return md_err_slot().c_str(); }

int mdhip_alloc(size_t nbytes, void **ptr_out) {
  size_t n = nbytes ? nbytes : 1;
  void *p = nullptr;
  if (posix_memalign(&p, 256, (n + 255) / 256 * 256) != 0 || !p)
    return md_fail(MDHIP_EMEMORY, "host alloc of %zu bytes failed", nbytes);
  std::lock_guard<std::mutex> lk(g_mu);
  g_live[p] = n;
  g_in_use += (int64_t)n;
  if (g_in_use > g_peak) g_peak = g_in_use;
  ++g_nalloc;
  *ptr_out = p;
  return MDHIP_OK;
}
int mdhip_free(void *p) {
  if (!p) return MDHIP_OK;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_live.find(p);
  if (it == g_live.end()) return md_fail(MDHIP_EVALUE, "free of unknown pointer %p", p);
  g_in_use -= (int64_t)it->second;
  g_live.erase(it);
  free(p);
  return MDHIP_OK;
}
int mdhip_empty_cache(void) { return MDHIP_OK; }
int mdhip_mem_stats(int64_t s[4]) {
  std::lock_guard<std::mutex> lk(g_mu);
  s[0] = g_in_use; s[1] = 0; s[2] = g_peak; s[3] = g_nalloc;
  return MDHIP_OK;
}
int mdhip_host_alloc(size_t n, void **p) { *p = malloc(n ? n : 1); return *p ? MDHIP_OK : md_fail(MDHIP_EMEMORY, "host_alloc failed"); }
int mdhip_host_free(void *p) { free(p); return MDHIP_OK; }
int mdhip_h2d(void *d, const void *s, size_t n) { if (n) memcpy(d, s, n); return MDHIP_OK; }
int mdhip_d2h(void *d, const void *s, size_t n) { if (n) memcpy(d, s, n); return MDHIP_OK; }
int mdhip_d2d(void *d, const void *s, size_t n) { if (n) memmove(d, s, n); return MDHIP_OK; }
int mdhip_sync(void) { return MDHIP_OK; }

struct HostEvent { std::chrono::steady_clock::time_point t; };
int mdhip_event_create(void **ev) { *ev = new HostEvent(); return MDHIP_OK; }
int mdhip_event_record(void *ev) { ((HostEvent *)ev)->t = std::chrono::steady_clock::now(); return MDHIP_OK; }
int mdhip_event_elapsed_ms(void *a, void *b, float *ms) {
  *ms = std::chrono::duration<float, std::milli>(((HostEvent *)b)->t - ((HostEvent *)a)->t).count();
  return MDHIP_OK;
}
int mdhip_event_destroy(void *ev) { delete (HostEvent *)ev; return MDHIP_OK; }
// kernel-attached timing: the double stamps the two events around the next f32 product it computes
static thread_local HostEvent *t_prof_start = nullptr, *t_prof_stop = nullptr;
int mdhip_event_attach_next(void *start, void *stop) {
  if (!start || !stop) return md_fail(MDHIP_EVALUE, "event_attach_next: two events");
  t_prof_start = (HostEvent *)start; t_prof_stop = (HostEvent *)stop;
  return MDHIP_OK;
}
int mdhip_event_attach_cancel(int *was_pending) {
  if (was_pending) *was_pending = t_prof_start != nullptr;
  t_prof_start = t_prof_stop = nullptr;
  return MDHIP_OK;
}
namespace {
struct ProfScope {   // (the device library attaches to f32 matrix-core kernels only; so does the double)
  HostEvent *e0 = nullptr, *e1 = nullptr;
  explicit ProfScope(bool f32) {
    if (f32 && t_prof_start) { e0 = t_prof_start; e1 = t_prof_stop; t_prof_start = t_prof_stop = nullptr; e0->t = std::chrono::steady_clock::now(); }
  }
  ~ProfScope() { if (e1) e1->t = std::chrono::steady_clock::now(); }
};
}  // namespace

// graphs: the double executes immediately, so "capture" records nothing and replay cannot
// re-run anything; it only accepts the call sequence (tests of the Python wrapper's state
// machine); numerical replay is covered on the GPU.
static int g_host_capturing = 0;
int mdhip_graph_begin(void) { if (g_host_capturing) return md_fail(MDHIP_ERUNTIME, "a capture is already in progress"); g_host_capturing = 1; return MDHIP_OK; }
int mdhip_graph_end(void **g) { if (!g_host_capturing) return md_fail(MDHIP_ERUNTIME, "no capture in progress"); g_host_capturing = 0; *g = nullptr; return md_fail(MDHIP_ERUNTIME, "the CPU test double cannot replay graphs"); }
int mdhip_graph_launch(void *) { return md_fail(MDHIP_ERUNTIME, "the CPU test double cannot replay graphs"); }
int mdhip_graph_destroy(void *) { return MDHIP_OK; }

static inline bool narrow_arr(const mdhip_array *a) { return a && !a->is_scalar && md_is_narrow(a->dtype); }
int mdhip_unary(int op, const mdhip_array *x, const mdhip_array *out) {
  if (op != MDHIP_U_COPY && x && out && (narrow_arr(x) || narrow_arr(out))) return md_narrow_unary_dispatch<HostExec>(op, x, out);
  return md_unary_dispatch<HostExec>(op, x, out);
}
int mdhip_convert(const mdhip_array *x, const mdhip_array *out) {
  MD_TRY(md_check_any_array(x, "convert x"));
  MD_TRY(md_check_any_array(out, "convert out"));
  MdIter it;
  const mdhip_array *ops[2] = {x, out};
  MD_TRY(md_build_iter(&it, 2, ops, out));
  int64_t offs[MD_MAX_OPS];
  for (int64_t i = 0; i < it.total; ++i) {
    md_iter_offsets(it, i, offs);
    switch (md_dtype_carrier(x->dtype)) {
      case 2: md_store_any<double>(out->data, out->dtype, offs[1], md_load_any<double>(x->data, x->dtype, offs[0])); break;
      case 1: md_store_any<uint64_t>(out->data, out->dtype, offs[1], md_load_any<uint64_t>(x->data, x->dtype, offs[0])); break;
      default: md_store_any<int64_t>(out->data, out->dtype, offs[1], md_load_any<int64_t>(x->data, x->dtype, offs[0])); break;
    }
  }
  return MDHIP_OK;
}
int mdhip_binary(int op, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out, int cdt) {
  if (a && b && out && (narrow_arr(a) || narrow_arr(b) || narrow_arr(out) || md_is_narrow(cdt))) return md_narrow_binary_dispatch<HostExec>(op, a, b, out, cdt);
  return md_binary_dispatch<HostExec>(op, a, b, out, cdt);
}
int mdhip_where(const mdhip_array *c, const mdhip_array *a, const mdhip_array *b, const mdhip_array *out) {
  if (c && a && b && out && (narrow_arr(c) || narrow_arr(a) || narrow_arr(b) || narrow_arr(out))) return md_narrow_where_dispatch<HostExec>(c, a, b, out);
  return md_where_dispatch<HostExec>(c, a, b, out);
}
int mdhip_fill(const mdhip_array *out, const mdhip_array *scalar) {
  if (!scalar || !scalar->is_scalar) return md_fail(MDHIP_EVALUE, "fill: value must be a scalar descriptor");
  return md_unary_dispatch<HostExec>(MDHIP_U_COPY, scalar, out);
}
int mdhip_arange(const mdhip_array *out, double start, double step) {
  MD_TRY(md_check_array(out, "arange out"));
  if (out->ndim != 1) return md_fail(MDHIP_EVALUE, "arange: out must be 1-D");
  for (int64_t i = 0; i < out->shape[0]; ++i) {
    int64_t o = i * out->strides[0];
    switch (out->dtype) {
      case MDHIP_I32: ((int32_t *)out->data)[o] = (int32_t)((int64_t)start + i * (int64_t)step); break;
      case MDHIP_I64: ((int64_t *)out->data)[o] = (int64_t)start + i * (int64_t)step; break;
      case MDHIP_F32: ((float *)out->data)[o] = (float)(start + (double)i * step); break;
      case MDHIP_F64: ((double *)out->data)[o] = start + (double)i * step; break;
      default: return md_fail(MDHIP_ETYPE, "arange: unsupported dtype");
    }
  }
  return MDHIP_OK;
}
// opt-in device RNG: the same Philox stream as the device kernels (md_rng.h), element by element
int mdhip_random_fill(int kind, uint64_t seed, uint64_t offset, double a, double b, const mdhip_array *out) {
  MD_TRY(md_check_array(out, "random out"));
  int64_t n = 1, expect = 1;
  for (int d = out->ndim - 1; d >= 0; --d) {
    if (out->shape[d] != 1 && out->strides[d] != expect) return md_fail(MDHIP_EVALUE, "random_fill: out must be C-contiguous");
    expect *= out->shape[d];
    n *= out->shape[d];
  }
  const bool f = out->dtype == MDHIP_F32 || out->dtype == MDHIP_F64, in = out->dtype == MDHIP_I32 || out->dtype == MDHIP_I64;
  if (kind < 0 || kind > 3) return md_fail(MDHIP_EVALUE, "random_fill: unknown kind %d", kind);
  if ((kind <= MD_RNG_NORMAL) ? !f : !in) return md_fail(MDHIP_ETYPE, "random_fill: kind %d cannot fill %s", kind, md_dtype_name(out->dtype));
  if (kind == MD_RNG_INTEGERS && (!(b >= 1.0) || b > 9007199254740992.0)) return md_fail(MDHIP_EVALUE, "random_fill: integers need 1 <= high - low <= 2^53");
  if (kind == MD_RNG_BINOMIAL && (!(a >= 0.0) || a > (double)MD_RNG_BINOMIAL_MAX_N || !(b >= 0.0 && b <= 1.0)))
    return md_fail(MDHIP_EVALUE, "random_fill: binomial needs 0 <= n <= %d and 0 <= p <= 1", MD_RNG_BINOMIAL_MAX_N);
  for (int64_t i = 0; i < n; ++i) {
    switch (kind) {
      case MD_RNG_UNIFORM:
        if (out->dtype == MDHIP_F32) ((float *)out->data)[i] = md_rng_uniform<float>(seed, offset, i);
        else ((double *)out->data)[i] = md_rng_uniform<double>(seed, offset, i);
        break;
      case MD_RNG_NORMAL:
        if (out->dtype == MDHIP_F32) ((float *)out->data)[i] = md_rng_normal<float>(seed, offset, i);
        else ((double *)out->data)[i] = md_rng_normal<double>(seed, offset, i);
        break;
      default: {
        const int64_t v = kind == MD_RNG_INTEGERS ? md_rng_integer(seed, offset, i, (int64_t)a, (uint64_t)b) : md_rng_binomial(seed, offset, i, (int64_t)a, b);
        if (out->dtype == MDHIP_I64) ((int64_t *)out->data)[i] = v;
        else ((int32_t *)out->data)[i] = (int32_t)v;
      }
    }
  }
  return MDHIP_OK;
}
int mdhip_random_permutation(uint64_t seed, uint64_t offset, const mdhip_array *out) {
  MD_TRY(md_check_array(out, "permutation out"));
  if (out->dtype != MDHIP_I64 || out->ndim != 1 || (out->shape[0] > 1 && out->strides[0] != 1))
    return md_fail(MDHIP_EVALUE, "random_permutation: out must be a contiguous 1-D int64 array");
  const int64_t n = out->shape[0];
  std::vector<uint64_t> keys((size_t)n);
  for (int64_t i = 0; i < n; ++i) keys[(size_t)i] = md_rng_key(seed, offset, i);
  int64_t *ids = (int64_t *)out->data;
  std::iota(ids, ids + n, (int64_t)0);
  std::stable_sort(ids, ids + n, [&](int64_t x, int64_t y) { return keys[(size_t)x] < keys[(size_t)y]; });   // (stable, like the device's LSD radix sort)
  return MDHIP_OK;
}
int mdhip_reduce(int op, const mdhip_array *x, const mdhip_array *out, uint32_t mask) {
  return md_reduce_any_out<HostExec>(op, x, out, mask);
}
// variance / std along one axis: NumPy's _var spelled out (mean by a true division, centred squares, division, sqrt), sequential sums.
// Same covered forms as the product (so that both the fused call and the composed fallback get exercised on the CPU too).
}  // extern "C"
template <class T> static void host_var(const T *x, int64_t outer, int64_t n, int64_t inner, T *out, T denom, int take_sqrt) {
  for (int64_t o = 0; o < outer; ++o)
    for (int64_t i = 0; i < inner; ++i) {
      const T *p = x + o * n * inner + i;
      T s = 0;
      for (int64_t k = 0; k < n; ++k) s += p[k * inner];
      const T mean = s / (T)n;
      T q = 0;
      for (int64_t k = 0; k < n; ++k) { const T d = p[k * inner] - mean; q += d * d; }
      const T v = q / denom;
      out[o * inner + i] = take_sqrt ? std::sqrt(v) : v;
    }
}
extern "C" {
int mdhip_var(const mdhip_array *x, const mdhip_array *out, int32_t axis, int64_t ddof, int take_sqrt) {
  MD_TRY(md_check_array(x, "var x"));
  MD_TRY(md_check_array(out, "var out"));
  if (x->is_scalar || out->is_scalar) return md_fail(MDHIP_EVALUE, "var: arrays expected");
  if (x->dtype != MDHIP_F32 && x->dtype != MDHIP_F64) return md_fail(MDHIP_EVALUE, "var: float32 / float64 only (the caller composes the rest)");
  if (out->dtype != x->dtype) return md_fail(MDHIP_ETYPE, "var: out dtype must equal x dtype");
  if (axis < 0 || axis >= x->ndim) return md_fail(MDHIP_EVALUE, "var: axis out of range");
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
  if (osz != outer * inner) return md_fail(MDHIP_EVALUE, "var: out has the wrong number of elements");
  const int64_t V = x->dtype == MDHIP_F32 ? 4 : 2;
  if (((uintptr_t)x->data & 15) || ((uintptr_t)out->data & 15)) return md_fail(MDHIP_EVALUE, "var: unaligned operands");
  if (inner == 1) {
    if (n % V) return md_fail(MDHIP_EVALUE, "var: row length not a multiple of the 16-B vector");
    // a row is one wave's / one block's work: a few very long rows would leave the chip idle (the composed passes use all of it)
    if (outer < 256 && n > 16384) return md_fail(MDHIP_EVALUE, "var: few long rows (the caller composes)");
  } else if (outer == 1) {
    if ((inner % V) || n < 64 || inner < 256) return md_fail(MDHIP_EVALUE, "var: column form needs >= 256 aligned columns and >= 64 rows");
  } else {
    return md_fail(MDHIP_EVALUE, "var: reduced axis in the middle (the caller composes)");
  }
  if (x->dtype == MDHIP_F32) host_var<float>((const float *)x->data, outer, n, inner, (float *)out->data, (float)(n - ddof), take_sqrt);
  else host_var<double>((const double *)x->data, outer, n, inner, (double *)out->data, (double)(n - ddof), take_sqrt);
  return MDHIP_OK;
}
int mdhip_matmul(const mdhip_array *a, const mdhip_array *b, const mdhip_array *c) {
  ProfScope prof(a && a->dtype == MDHIP_F32);
  return md_matmul_dispatch<HostExec>(a, b, c);
}
// fused GEMM + bias + relu-sum + mask: plain loops here (k-ordered float fma chain per element, as the MFMA kernel)
int mdhip_matmul_bias_relu_sum(const mdhip_array *a, const mdhip_array *b, const mdhip_array *bias,
                               const mdhip_array *mask_out, const mdhip_array *sum_out) {
  MD_TRY(md_check_array(a, "matmul a"));
  MD_TRY(md_check_array(b, "matmul b"));
  MD_TRY(md_check_array(bias, "bias"));
  MD_TRY(md_check_array(mask_out, "mask"));
  MD_TRY(md_check_array(sum_out, "sum"));
  if (a->dtype != MDHIP_F32 || b->dtype != MDHIP_F32 || bias->dtype != MDHIP_F32 || sum_out->dtype != MDHIP_F32 || mask_out->dtype != MDHIP_BOOL)
    return md_fail(MDHIP_ETYPE, "matmul_bias_relu_sum: float32 operands, bool mask, float32 sum");
  if (a->ndim != 2 || b->ndim != 2 || mask_out->ndim != 2 || bias->ndim != 1)
    return md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: 2-D operands and mask, 1-D bias");
  const int64_t M = a->shape[0], K = a->shape[1], N = b->shape[1];
  if (b->shape[0] != K || bias->shape[0] != N || mask_out->shape[0] != M || mask_out->shape[1] != N)
    return md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: shapes do not agree");
  if ((M % 64) || (N % 64) || (K % 16))   // (the double keeps the product's notion of "covered", so that both paths get exercised)
    return md_fail(MDHIP_EVALUE, "matmul_bias_relu_sum: shape not covered by the fused kernel");
  ProfScope prof(true);
  const float *A = (const float *)a->data, *B = (const float *)b->data, *bv = (const float *)bias->data;
  uint8_t *mk = (uint8_t *)mask_out->data;
  float total = 0.0f;
  for (int64_t m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) {
      float acc = 0.0f;
      for (int64_t k = 0; k < K; ++k) acc = fmaf(A[m * a->strides[0] + k * a->strides[1]], B[k * b->strides[0] + n * b->strides[1]], acc);
      const float z = acc + bv[n * bias->strides[0]];
      mk[m * mask_out->strides[0] + n * mask_out->strides[1]] = z > 0.0f;
      total += z > 0.0f ? z : 0.0f;
    }
  ((float *)sum_out->data)[0] = total;
  return MDHIP_OK;
}

int mdhip_gather(const mdhip_index_plan *pl, const void *src, int dtype, const mdhip_array *out) {
  MD_TRY(md_check_plan(pl));
  MD_TRY(md_check_any_array(out, "gather out"));
  if (dtype < 0 || dtype >= MDHIP_NUM_ALL_DTYPES || out->dtype != dtype) return md_fail(MDHIP_ETYPE, "gather: bad dtype code %d (out has %d)", dtype, out->dtype);
  if (out->ndim != pl->ndim) return md_fail(MDHIP_EVALUE, "gather: out ndim mismatch");
  size_t es = md_dtype_size(dtype);
  int64_t n = md_plan_total(pl), pos[MDHIP_MAX_NDIM];
  for (int64_t i = 0; i < n; ++i) {
    bool oob = false;
    int64_t off = md_plan_offset(*pl, i, pos, &oob);
    if (oob) return md_fail(MDHIP_EINDEX, "index out of bounds");
    int64_t oo = 0;
    for (int d = 0; d < pl->ndim; ++d) oo += pos[d] * out->strides[d];
    memcpy((char *)out->data + oo * es, (const char *)src + off * es, es);
  }
  return MDHIP_OK;
}
int mdhip_scatter(const mdhip_index_plan *pl, void *dst, int dtype, const mdhip_array *val, int mode) {
  MD_TRY(md_check_plan(pl));
  if (!val) return md_fail(MDHIP_EVALUE, "scatter: null value");
  if (!val->is_scalar) MD_TRY(md_check_any_array(val, "scatter val"));
  if (!val->is_scalar && val->dtype != dtype) return md_fail(MDHIP_ETYPE, "scatter: value dtype must match destination");
  if (!val->is_scalar && val->ndim != pl->ndim) return md_fail(MDHIP_EVALUE, "scatter: value ndim mismatch");
  if (mode != MDHIP_SCATTER_SET && mode != MDHIP_SCATTER_ADD) return md_fail(MDHIP_EVALUE, "scatter: bad mode %d", mode);
  if (dtype < 0 || dtype >= MDHIP_NUM_ALL_DTYPES) return md_fail(MDHIP_ETYPE, "scatter: bad dtype code %d", dtype);
  size_t es = md_dtype_size(dtype);
  int64_t n = md_plan_total(pl), pos[MDHIP_MAX_NDIM];
  for (int64_t i = 0; i < n; ++i) {  // bounds first: NumPy raises before writing
    bool oob = false;
    md_plan_offset(*pl, i, pos, &oob);
    if (oob) return md_fail(MDHIP_EINDEX, "index out of bounds");
  }
  for (int64_t i = 0; i < n; ++i) {
    bool oob = false;
    int64_t off = md_plan_offset(*pl, i, pos, &oob);
    int64_t vo = 0;
    if (!val->is_scalar) for (int d = 0; d < pl->ndim; ++d) vo += pos[d] * val->strides[d];
#define MD_SC(code, T)                                                                    \
  case code: {                                                                            \
    T v = val->is_scalar ? md_scalar_as<T>(val) : ((const T *)val->data)[vo];             \
    T *d = (T *)dst + off;                                                                \
    if (mode == MDHIP_SCATTER_ADD) *d = md_storage_add(*d, v); else *d = v;               \
  } break;
    switch (dtype) {
      case MDHIP_BOOL: {
        uint8_t v = val->is_scalar ? md_scalar_as<uint8_t>(val) : ((const uint8_t *)val->data)[vo];
        uint8_t *d = (uint8_t *)dst + off;
        if (mode == MDHIP_SCATTER_ADD) *d = (uint8_t)(*d || v); else *d = v;
      } break;
      MD_SC(MDHIP_I32, int32_t)
      MD_SC(MDHIP_I64, int64_t)
      MD_SC(MDHIP_F32, float)
      MD_SC(MDHIP_F64, double)
      MD_SC(MDHIP_U32, int32_t)
      MD_SC(MDHIP_U64, int64_t)
      MD_SC(MDHIP_I8, int8_t)
      MD_SC(MDHIP_U8, int8_t)
      MD_SC(MDHIP_I16, int16_t)
      MD_SC(MDHIP_U16, int16_t)
      MD_SC(MDHIP_F16, f16)
    }
#undef MD_SC
    (void)es;
  }
  return MDHIP_OK;
}

// fused expressions: same interpreter (csrc/md_vm.h), one element at a time
extern "C++" {
template <class T> struct HostVmLoader {
  const mdhip_vm_program *pr;
  const int64_t *offs;
  void operator()(int l, T (&d)[1]) const { d[0] = md_load<T>(pr->leaves[l].data, pr->leaves[l].dtype, offs[l]); }
};
template <class T> static T host_vm_at(const mdhip_vm_program *pr, const MdVmIter &it, int64_t i, int64_t *out_off) {
  int64_t offs[MDHIP_VM_MAX_LEAVES + 1] = {0};
  int64_t lin = i;
  for (int d = it.ndim - 1; d >= 0; --d) {
    const int64_t e = it.shape[d], q = lin / e, r = lin - q * e;
    lin = q;
    for (int l = 0; l < pr->n_leaves; ++l) offs[l] += r * it.strides[l][d];
    offs[MDHIP_VM_MAX_LEAVES] += r * it.strides[MDHIP_VM_MAX_LEAVES][d];
  }
  HostVmLoader<T> ld{pr, offs};
  T r[1];
  MdVmFetchDirect fetch{pr->ctrl, pr->imm};
  md_vm_run<T, 1>(pr->n_instr, fetch, ld, r);
  if (out_off) *out_off = offs[MDHIP_VM_MAX_LEAVES];
  return r[0];
}
template <class T> static int host_vm_eval(const mdhip_vm_program *pr, const mdhip_array *out) {
  MdVmIter it;
  MD_TRY(md_vm_build_iter(&it, pr, out, out));
  for (int64_t i = 0; i < it.total; ++i) {
    int64_t oo;
    T v = host_vm_at<T>(pr, it, i, &oo);
    if (out->dtype == MDHIP_BOOL) ((uint8_t *)out->data)[oo] = (uint8_t)(v != (T)0);
    else ((T *)out->data)[oo] = v;
  }
  return MDHIP_OK;
}
template <class R, class T>
static int host_vm_reduce(const mdhip_vm_program *pr, const mdhip_array *shape_like, const mdhip_array *out, uint32_t mask) {
  const int nd = shape_like->ndim;
  const uint32_t all = nd ? ((1u << nd) - 1u) : 0u;
  // un-collapsed walk so (row, col) stay identifiable
  mdhip_array fake = *shape_like;
  MdVmIter it;
  MD_TRY(md_vm_build_iter(&it, pr, &fake, nullptr));
  if (it.total == 0) return md_fail(MDHIP_EVALUE, "vm_reduce: empty operand");
  if (mask == all) {
    T acc = R::template identity<T>();
    for (int64_t i = 0; i < it.total; ++i) acc = R::combine(acc, host_vm_at<T>(pr, it, i, nullptr));
    ((T *)out->data)[0] = acc;
    return MDHIP_OK;
  }
  if (nd == 2 && mask == 1u) {
    const int64_t rows = shape_like->shape[0], cols = shape_like->shape[1];
    // (a program whose operands are all dense or fully broadcast collapses to 1-D: the linear index still is r*cols + c)
    const bool flat = it.ndim == 1 && it.shape[0] == rows * cols;
    if (!(flat || (it.ndim == 2 && it.shape[0] == rows && it.shape[1] == cols)) || (cols & 3))
      return md_fail(MDHIP_EVALUE, "vm_reduce: geometry not supported");
    for (int64_t c = 0; c < cols; ++c) {
      T acc = R::template identity<T>();
      for (int64_t r = 0; r < rows; ++r) acc = R::combine(acc, host_vm_at<T>(pr, it, r * cols + c, nullptr));
      ((T *)out->data)[c] = acc;
    }
    return MDHIP_OK;
  }
  return md_fail(MDHIP_EVALUE, "vm_reduce: only full reductions and axis-0 reductions of 2-D programs are fused");
}
}  // extern "C++"
int mdhip_vm_eval(const mdhip_vm_program *pr, const mdhip_array *out) {
  MD_TRY(md_vm_check(pr));
  MD_TRY(md_check_array(out, "vm out"));
  if (out->dtype != pr->compute_dtype && out->dtype != MDHIP_BOOL)
    return md_fail(MDHIP_ETYPE, "vm_eval: out dtype must be the compute dtype or bool");
  return pr->compute_dtype == MDHIP_F32 ? host_vm_eval<float>(pr, out) : host_vm_eval<double>(pr, out);
}
int mdhip_vm_reduce(const mdhip_vm_program *pr, int op, const mdhip_array *shape_like, const mdhip_array *out, uint32_t mask) {
  MD_TRY(md_vm_check(pr));
  MD_TRY(md_check_array(shape_like, "vm shape"));
  MD_TRY(md_check_array(out, "vm out"));
  if (out->dtype != pr->compute_dtype) return md_fail(MDHIP_ETYPE, "vm_reduce: out dtype must be the compute dtype");
#define MD_VMR(R) return pr->compute_dtype == MDHIP_F32 ? host_vm_reduce<R, float>(pr, shape_like, out, mask) : host_vm_reduce<R, double>(pr, shape_like, out, mask)
  switch (op) {
    case MDHIP_R_SUM: MD_VMR(RSum);
    case MDHIP_R_PROD: MD_VMR(RProd);
    case MDHIP_R_MAX: MD_VMR(RMax);
    case MDHIP_R_MIN: MD_VMR(RMin);
  }
#undef MD_VMR
  return md_fail(MDHIP_EVALUE, "vm_reduce: reduce op %d is not fused", op);
}

static int64_t host_nz(const mdhip_array *x, int64_t *out) {
  int64_t n = 1;
  for (int d = 0; d < x->ndim; ++d) n *= x->shape[d];
  int64_t c = 0;
  for (int64_t i = 0; i < n; ++i)
    if (md_load<uint8_t>(x->data, x->dtype, i)) { if (out) out[c] = i; ++c; }
  return c;
}
int mdhip_nonzero_count(const mdhip_array *x, int64_t *count_out) { *count_out = host_nz(x, nullptr); return MDHIP_OK; }
int mdhip_nonzero_fill(const mdhip_array *x, int64_t, int64_t *out_flat) { host_nz(x, out_flat); return MDHIP_OK; }

int mdhip_vm_jit_probe(const mdhip_vm_program *, int, int, int, char *, size_t) {
  return md_fail(MDHIP_ERUNTIME, "the CPU test double has no run-time compiler");
}
int mdhip_vm_jit_stats(int64_t stats[2]) { stats[0] = stats[1] = 0; return MDHIP_OK; }
int mdhip_vm_eval_multi(const mdhip_vm_program *progs, const mdhip_array *outs, int n) {
  if (n < 1) return md_fail(MDHIP_EVALUE, "vm_eval_multi: no programs");
  for (int k = 0; k < n; ++k) MD_TRY(mdhip_vm_eval(&progs[k], &outs[k]));
  return MDHIP_OK;
}
int mdhip_vm_jit_probe_multi(const mdhip_vm_program *, int, char *, size_t) {
  return md_fail(MDHIP_ERUNTIME, "the CPU test double has no run-time compiler");
}
// one-pass eval + column reduce: here simply the two plain loops, same results
int mdhip_vm_eval_reduce_cols(const mdhip_vm_program *pr, int op, const mdhip_array *out_eval, const mdhip_array *out_red) {
  MD_TRY(md_vm_check(pr));
  MD_TRY(md_check_array(out_eval, "vm out"));
  MD_TRY(md_check_array(out_red, "vm out"));
  if (out_eval->dtype != pr->compute_dtype || out_red->dtype != pr->compute_dtype)
    return md_fail(MDHIP_ETYPE, "vm_eval_reduce_cols: both outputs must have the compute dtype");
  if (out_eval->ndim != 2 || out_eval->shape[0] < 2 || (out_eval->shape[1] & 3))
    return md_fail(MDHIP_EVALUE, "vm_eval_reduce_cols: shape not covered by the one-pass kernel");
  MD_TRY(mdhip_vm_eval(pr, out_eval));
  return mdhip_vm_reduce(pr, op, out_eval, out_red, 1u);
}

// data-parallel entry points: the double has no collective; world size 1 only.
static int g_nranks = 0;
int mdhip_comm_probe(void) { return MDHIP_OK; }
int mdhip_comm_count(int *n) {
  if (g_nranks != 1) return md_fail(MDHIP_ERUNTIME, "communicator not initialised");
  *n = g_nranks;
  return MDHIP_OK;
}
int mdhip_comm_get_unique_id(uint8_t uid[MDHIP_UID_BYTES]) { memset(uid, 0, MDHIP_UID_BYTES); return MDHIP_OK; }
int mdhip_comm_init(int nranks, int, const uint8_t *) {
  if (nranks != 1) return md_fail(MDHIP_ERUNTIME, "host test double has no collective backend (nranks=%d)", nranks);
  g_nranks = 1;
  return MDHIP_OK;
}
int mdhip_comm_allreduce_sum(void *, size_t, int) {
  if (g_nranks != 1) return md_fail(MDHIP_ERUNTIME, "communicator not initialised");
  return MDHIP_OK;
}
int mdhip_comm_allreduce_sum_async(void *b, size_t n, int d) { return mdhip_comm_allreduce_sum(b, n, d); }
int mdhip_comm_wait(void) { return MDHIP_OK; }
int mdhip_comm_destroy(void) { g_nranks = 0; return MDHIP_OK; }

}  // extern "C"
