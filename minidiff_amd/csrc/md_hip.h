// md_hip.h — device-side helpers shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>

#include "md_dispatch.h"

#include "md_ticket.h"

hipStream_t md_stream();
struct MdGemm;
int md_gemm_skinny(const MdGemm &g, int dtype);           // skinny.hip: thin products (matrix x vector ..); -1 = not applicable
int md_gemm_longk(const MdGemm &g, int dtype);            // skinny.hip: both sides thin, k long (a dot product ..); -1 = not applicable
unsigned *md_tickets();                                   // MD_TICKET_WORDS zeroed counters (md_ticket.h)
bool md_capturing();                                      // a stream capture is recording (mdhip_graph_begin .. _end)
int *md_sticky();                                         // host-mapped word a CAPTURED gather / scatter sets on an out-of-bounds index
int md_sticky_check();                                    // after a synchronisation: MDHIP_EINDEX (and clear) if a replay set it
bool md_prof_take(hipEvent_t *start, hipEvent_t *stop);   // bench timing: events for the next main kernel (MD_LAUNCH, the GEMM launchers), if any were attached
int md_hip_check(hipError_t e, const char *what);

#define MD_LAUNCH_CHECK(name) md_hip_check(hipGetLastError(), name)

// Launch on the library's stream. With events attached (mdhip_event_attach_next: bench.py) the dispatch ITSELF carries the
// start / stop timestamps — the kernel's own duration, no marker packets around it (a marker pair costs ~4 us of stream
// time and reads 5-15 % low on a 25-us streaming kernel). Kernel names with template commas go in parentheses.
#define MD_LAUNCH(kernel, grid, block, ...)                                                                            \
  do {                                                                                                                 \
    hipEvent_t md_e0_, md_e1_;                                                                                         \
    if (md_prof_take(&md_e0_, &md_e1_)) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, md_stream(), md_e0_, md_e1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, md_stream(), __VA_ARGS__);                             \
  } while (0)

// MI355X: 256 CUs. Streaming kernels cap the grid and stride (guide §6 G11).
constexpr int MD_NUM_CUS = 256;
constexpr int MD_BLOCK = 256;
static inline int md_max_blocks() {
  const int n = (int)md_opt(MD_OPT_MAX_BLOCKS);  // experiment knob (md_options.h)
  return n > 0 ? n : MD_NUM_CUS * 8;
}
static inline int md_grid_for(int64_t work_items, int per_block = MD_BLOCK, int max_blocks = 0) {
  if (max_blocks <= 0) max_blocks = md_max_blocks();
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

template <class T, int N> struct alignas(sizeof(T) * N > 16 ? 16 : sizeof(T) * N) MdVec {
  T v[N];
};

// ---- wave / block reductions (wave = 64 lanes) ----------------------------------
template <class T> __device__ __forceinline__ T md_shfl_down(T v, int delta) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int32_t w[2]; } u;
    u.t = v;
    u.w[0] = __shfl_down(u.w[0], delta, 64);
    u.w[1] = __shfl_down(u.w[1], delta, 64);
    return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int32_t w; } u;
    u.t = v;
    u.w = __shfl_down(u.w, delta, 64);
    return u.t;
  } else {
    int32_t w = (int32_t)v;
    w = __shfl_down(w, delta, 64);
    return (T)w;
  }
}
