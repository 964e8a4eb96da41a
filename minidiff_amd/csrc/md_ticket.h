// md_ticket.h — hand-off INSIDE one launch for the split reductions: every block publishes its partial result, the block
// that arrives LAST at a counter combines all of them (in index order: the result does not depend on who was last).
// Replaces the second ("merge" / "finish") launch of reduce.hip, fusion.hip, fusion_jit.inc and the GEMM epilogue:
// 4.5-4.8 us of kernel + 1.7 us of launch boundary for a few KiB of partials (profiles/r3_reduce_lab.txt).
//
// Protocol (CDNA guide §6 Guideline 16, form R1 with a returned counter add; MI355X_MICROARCH "visibility", first table row):
//   producer : partials stored WRITE-THROUGH (sc1: buffer store aux 16, or a relaxed agent-scope atomic store), every
//              storing wave drains them (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds 1 to the counter
//              (relaxed, agent scope, returning).
//   consumer : the block whose add returned n - 1. Its other waves learn it through LDS behind a barrier; EVERY load of
//              a partial is an sc1 load (L1 bypassed: a line of the partial buffer may sit in this CU's L1 from the
//              previous call, the buffer addresses repeat) — no acquire fence needed, none used.
//   counters : live in one process-wide block (md_tickets(), runtime.hip), zero between launches: whoever completes a
//              count stores 0 back. All launches that use them are ordered on libmdhip's one compute stream.
// Plain builtins only: this text is also compiled by hiprtc in front of the generated fused kernels.
#pragma once

#define MD_TICKET_WORDS 16384  // 64 KiB of counters
#define MD_TICKET_PAD 16       // a counter that many blocks hit gets a 64-B line of its own

typedef int md_i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t md_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
// write-through store / L1-bypassing load of a value of 16 or 32 bytes (one or two 16-B buffer operations)
template <class V> __device__ __forceinline__ void md_st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const V &v) {
  static_assert(sizeof(V) % 16 == 0, "multiples of 16 bytes only");
  md_i32x4 t[sizeof(V) / 16];
  __builtin_memcpy(t, &v, sizeof(V));
#pragma unroll
  for (unsigned i = 0; i < sizeof(V) / 16; ++i) __builtin_amdgcn_raw_buffer_store_b128(t[i], r, byte_off + 16u * i, 0, 16);
}
template <class V> __device__ __forceinline__ V md_ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  static_assert(sizeof(V) % 16 == 0, "multiples of 16 bytes only");
  md_i32x4 t[sizeof(V) / 16];
#pragma unroll
  for (unsigned i = 0; i < sizeof(V) / 16; ++i) t[i] = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 16u * i, 0, 16);
  V v;
  __builtin_memcpy(&v, t, sizeof(V));
  return v;
}
// 4- / 8-byte scalars (float, double, int32, int64)
template <class T> __device__ __forceinline__ void md_st_sc1(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T> __device__ __forceinline__ T md_ld_sc1(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The last block's read of n scalar partials: thread t folds partials t, t + blockDim, .. in index order, eight loads in
// flight (issued before the first combine: one round trip per eight, not one per partial). R: a reducer of md_ops.h.
template <class R, class T> __device__ __forceinline__ T md_fold_partials(const T *partial, unsigned n) {
  T a = R::template identity<T>();
  for (unsigned base = threadIdx.x; base < n; base += 8u * blockDim.x) {
    T v[8];
#pragma unroll
    for (unsigned u = 0; u < 8u; ++u) {
      const unsigned i = base + u * blockDim.x;
      v[u] = i < n ? md_ld_sc1(partial + i) : R::template identity<T>();
    }
#pragma unroll
    for (unsigned u = 0; u < 8u; ++u) a = R::combine(a, v[u]);
  }
  return a;
}

// Call from EVERY thread of the block, after the block's partial stores. True in all threads of the block that arrived last.
__device__ __forceinline__ bool md_ticket_last(unsigned *ticket, unsigned n_arrivals, unsigned *lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = old == n_arrivals - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0u;
}
// Many arrivals (a full reduction's ~1000 blocks): adds to ONE word serialise at ~12 ns each, so arrivals go to 32 shard
// counters (arrival i -> shard i % 32, each on a line of its own) and the last arrival of a shard to a top counter.
// Words used: t[0] (top) and t[MD_TICKET_PAD * (1 + shard)].
#define MD_TICKET2_WORDS (MD_TICKET_PAD * 33)
__device__ __forceinline__ bool md_ticket_last2(unsigned *t, unsigned idx, unsigned n_arrivals, unsigned *lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned last = 0u;
    const unsigned sh = idx & 31u, cnt = (n_arrivals - sh + 31u) >> 5, shards = n_arrivals < 32u ? n_arrivals : 32u;
    unsigned *c = t + MD_TICKET_PAD * (1u + sh);
    if (__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == cnt - 1u) {
      __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1u) {
        __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = 1u;
      }
    }
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0u;
}
