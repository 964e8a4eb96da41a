/*
 * mdhip.h — C-ABI of libmdhip.so, the MI355X (gfx950) array runtime that sits
 * behind minidiff's backend boundary.
 *
 * Every entry point below replaces one group of names in the reference's
 * backend function table (reference: minidiff/backend/numpy.py:14-206, the
 * alias table onto NumPy; abstract surface minidiff/backend/__init__.py:88-759).
 * The reference has no FFI of its own: its "binding" is a Python class whose
 * attributes are array functions, so the C boundary is what a ctypes shim
 * (minidiff_amd/hip_backend.py) needs in order to serve those attributes.
 * INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *  - plain C: pointers, sizes, PODs; no C++ / torch types cross this line.
 *  - every function returns an int status (MDHIP_OK or an MDHIP_E* code that
 *    the shim maps onto ValueError / TypeError / IndexError / MemoryError /
 *    RuntimeError, mirroring the exception types NumPy raises for the
 *    reference); mdhip_last_error() returns the thread-local message.
 *  - all device work is enqueued on ONE stream per process (the reference is
 *    single-threaded eager: stream order == program order); only the calls
 *    documented as synchronising wait for the device.
 *  - arrays are described by value with mdhip_array: a device pointer to the
 *    element at index (0,...,0), a dtype code, and shape/strides in ELEMENTS
 *    (strides may be 0 for broadcast views or negative for flipped views).
 *  - Python scalars travel as mdhip_array with is_scalar != 0 (no device
 *    memory; value in scalar_f / scalar_i), which keeps NumPy's weak-scalar
 *    promotion in the caller and the arithmetic on the device.
 */
#ifndef MDHIP_H
#define MDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDHIP_MAX_NDIM 8

/* ---- status codes -------------------------------------------------------- */
enum {
  MDHIP_OK = 0,
  MDHIP_EVALUE = 1,   /* -> ValueError  */
  MDHIP_ETYPE = 2,    /* -> TypeError   */
  MDHIP_EINDEX = 3,   /* -> IndexError  */
  MDHIP_EMEMORY = 4,  /* -> MemoryError */
  MDHIP_ERUNTIME = 5  /* -> RuntimeError (HIP / RCCL failure) */
};

/* ---- dtypes (reference: minidiff/backend/numpy.py:188-200) --------------- */
enum {
  MDHIP_BOOL = 0, /* 1 byte, values 0/1 (numpy.bool_) */
  MDHIP_I32 = 1,
  MDHIP_I64 = 2,
  MDHIP_F32 = 3,
  MDHIP_F64 = 4,
  MDHIP_NUM_DTYPES = 5, /* the COMPUTE dtypes: every arithmetic entry point takes these and only these */
  /* Storage-only dtypes — the other names of the reference table (numpy.py:188-200: float16, uint8/16/32/64, int8/16).
   * Arrays of these types live in device memory and are moved by mdhip_h2d / mdhip_d2h and converted by mdhip_convert;
   * arithmetic on them is promote -> compute in a wide type -> demote, decided on the host (what NumPy's own float16
   * loops do; integer wrap-around falls out of the truncating conversion). */
  MDHIP_I8 = 5,
  MDHIP_I16 = 6,
  MDHIP_U8 = 7,
  MDHIP_U16 = 8,
  MDHIP_U32 = 9,
  MDHIP_U64 = 10,
  MDHIP_F16 = 11, /* IEEE binary16 */
  MDHIP_NUM_ALL_DTYPES = 12
};

typedef struct mdhip_array {
  void *data;                       /* device pointer (NULL when is_scalar) */
  int32_t dtype;                    /* MDHIP_* dtype code */
  int32_t ndim;                     /* 0..MDHIP_MAX_NDIM */
  int64_t shape[MDHIP_MAX_NDIM];
  int64_t strides[MDHIP_MAX_NDIM];  /* in elements */
  int32_t is_scalar;                /* host scalar operand */
  int32_t _pad;
  int64_t scalar_i;                 /* value when dtype is BOOL/I32/I64 */
  double scalar_f;                  /* value when dtype is F32/F64 */
} mdhip_array;

/* ---- unary ops (reference: numpy.py:19-59; definitions.py:266-420) ------- */
enum {
  MDHIP_U_COPY = 0, /* copy / astype: out dtype may differ from in dtype */
  MDHIP_U_ABS,
  MDHIP_U_NEG,
  MDHIP_U_SIGN,
  MDHIP_U_CEIL,
  MDHIP_U_FLOOR,
  MDHIP_U_SIN,
  MDHIP_U_COS,
  MDHIP_U_TAN,
  MDHIP_U_SINH,
  MDHIP_U_COSH,
  MDHIP_U_TANH,
  MDHIP_U_EXP,
  MDHIP_U_LOG,
  MDHIP_U_SQRT,
  MDHIP_U_LOGICAL_NOT,
  MDHIP_U_INVERT,
  MDHIP_U_ISNAN,
  MDHIP_U_COUNT
};

/* ---- binary ops (reference: numpy.py:61-92; definitions.py:424-536) ------ */
enum {
  MDHIP_B_ADD = 0,
  MDHIP_B_SUB,
  MDHIP_B_MUL,
  MDHIP_B_TRUE_DIV,
  MDHIP_B_FLOOR_DIV,
  MDHIP_B_MOD,
  MDHIP_B_POW,
  MDHIP_B_MAXIMUM,
  MDHIP_B_MINIMUM,
  MDHIP_B_EQ,
  MDHIP_B_NE,
  MDHIP_B_LT,
  MDHIP_B_LE,
  MDHIP_B_GT,
  MDHIP_B_GE,
  MDHIP_B_LAND,
  MDHIP_B_LOR,
  MDHIP_B_LXOR,
  MDHIP_B_COUNT
};

/* ---- reductions (reference: numpy.py:20-23,43-46,56-57) ------------------ */
enum {
  MDHIP_R_SUM = 0,
  MDHIP_R_PROD,
  MDHIP_R_MAX,
  MDHIP_R_MIN,
  MDHIP_R_ANY,
  MDHIP_R_ALL,
  MDHIP_R_ARGMAX,
  MDHIP_R_ARGMIN,
  MDHIP_R_COUNT
};

/* ======================= runtime ========================================== */
/* Bind this process to HIP device `device`, create the stream and the caching
 * allocator. Idempotent for the same device. */
int mdhip_init(int device);
int mdhip_device(int *device_out);
/* Undo mdhip_init (stream, ticket block, cached device blocks). MDHIP_ERUNTIME while device blocks are still live. */
int mdhip_shutdown(void);
/* TEST / A-B HOOK. libmdhip's tuning switches (tile forcing, staging scheme, legacy two-launch reductions, ...) live in one
 * table (csrc/md_options.h). Without MDHIP_EXPERIMENTS=1 in the environment the library ignores every MDHIP_GEMM_* /
 * MDHIP_*_TICKET / MDHIP_COLS_* / MDHIP_ARG_* / MDHIP_SWEEP_* variable and never calls getenv on a launch path; tests that
 * must force a kernel set the table entry through this call (name = the entry's lower-case name, e.g. "gemm_cfg"), effective
 * from the next launch. Unknown name: MDHIP_EVALUE. No reference counterpart (the reference has no native code). */
int mdhip_debug_set_option(const char *name, int64_t value);
int mdhip_debug_get_option(const char *name, int64_t *value_out);
/* "hip:gfx950" for the product library; the CPU test double under oracle/
 * answers "host". The shim refuses to run product code on anything else. */
const char *mdhip_target(void);
const char *mdhip_last_error(void);
/* Device-memory blocks come from a size-binned caching allocator (the tape
 * frees temporaries by Python refcount: SURVEY.md §5 "Memory management"). */
int mdhip_alloc(size_t nbytes, void **ptr_out);
int mdhip_free(void *ptr);
int mdhip_empty_cache(void);
/* stats[0]=bytes in use, [1]=bytes cached, [2]=peak in use, [3]=#hipMalloc */
int mdhip_mem_stats(int64_t stats[4]);
int mdhip_h2d(void *dst, const void *src, size_t nbytes); /* async on stream for pinned src; ordered */
int mdhip_d2h(void *dst, const void *src, size_t nbytes); /* SYNCHRONISES */
int mdhip_d2d(void *dst, const void *src, size_t nbytes); /* async */
/* Page-locked host blocks from a size-binned cache, for the host side of large transfers
 * (as_numpy / array, numpy.py:174-185,204-206): a D2H copy into pageable, never-touched memory
 * runs at the page-fault rate (2-11 GB/s measured), into a pinned block at the link rate. At most
 * `MDHIP_PINNED_CAP` bytes (default 4 GiB) are outstanding; beyond that MDHIP_EMEMORY, and the
 * caller uses ordinary memory. */
int mdhip_host_alloc(size_t nbytes, void **ptr_out);
int mdhip_host_free(void *ptr);
int mdhip_sync(void);                                     /* stream + device */
/* HIP events on the library's stream, for bench.py's per-kernel timing. */
int mdhip_event_create(void **ev_out);
int mdhip_event_record(void *ev);
int mdhip_event_elapsed_ms(void *start, void *stop, float *ms_out); /* SYNCHRONISES on stop */
int mdhip_event_destroy(void *ev);
/* Kernel-attached timing (bench.py): the NEXT f32 matrix-core GEMM kernel this thread launches (mdhip_matmul,
 * mdhip_matmul_bias_relu_sum) records `start` when it begins and `stop` when it ends — timestamps of the dispatch itself
 * (hipExtLaunchKernel), with no marker packets between kernels (a mdhip_event_record bracket costs ~5 us of stream
 * time per marker). mdhip_event_attach_cancel drops a pending attachment and says whether there was one: a call that
 * launched no such kernel (deferred, split-K, float64, integer) left it pending, and the events were never recorded. */
int mdhip_event_attach_next(void *start, void *stop);
int mdhip_event_attach_cancel(int *was_pending_out);

/* ======================= hipGraph capture / replay ========================== */
/* Capture everything enqueued on the library's stream between begin and end into a
 * hipGraph and replay it with one launch (removes per-call host dispatch and launch
 * gaps for repeated sweeps of small graphs). Device blocks allocated
 * while capturing stay RESERVED for the graph (temporaries are recycled only inside
 * it, results keep their addresses) until mdhip_graph_destroy, so a replay rewrites
 * the very arrays the captured run returned. Calls that must synchronise (D2H,
 * data-dependent sizes, index bounds checks, H2D uploads) are not capturable and fail
 * with MDHIP_ERUNTIME; mdhip_graph_end must still be called (it aborts the capture). */
int mdhip_graph_begin(void);
int mdhip_graph_end(void **graph_out);
int mdhip_graph_launch(void *graph);
int mdhip_graph_destroy(void *graph);

/* ======================= elementwise ====================================== */
/* out = op(x). `out` describes freshly allocated (or in-place) memory with the
 * result dtype chosen by the caller from NumPy's type resolution. */
int mdhip_unary(int op, const mdhip_array *x, const mdhip_array *out);
/* out = x converted element by element (C conversion rules = numpy.ndarray.astype's unsafe casting; to bool: x != 0;
 * binary16 <-> wider floats with round-to-nearest-even), ANY pair of the 12 dtypes above, any strides (<= 8-D): the one
 * kernel behind the storage-only dtypes — `astype`, strided copies, and the promote / demote steps around wide arithmetic.
 * Replaces numpy.py:188-200's dtypes as used through `astype` (reference: minidiff/tensor.py:105, backend/numpy.py:19). */
int mdhip_convert(const mdhip_array *x, const mdhip_array *out);
/* out = op(a, b) with NumPy broadcasting already applied by the caller:
 * a, b and out have the same ndim/shape; broadcast axes carry stride 0.
 * compute_dtype = the ufunc loop dtype NumPy resolves (np.<op>.resolve_dtypes). */
int mdhip_binary(int op, const mdhip_array *a, const mdhip_array *b,
                 const mdhip_array *out, int compute_dtype);
/* out = cond ? a : b            (reference: numpy.py:95, definitions.py:555-559) */
int mdhip_where(const mdhip_array *cond, const mdhip_array *a, const mdhip_array *b,
                const mdhip_array *out);
/* fill out with a scalar (ones/zeros/full families, numpy.py:98-103) */
int mdhip_fill(const mdhip_array *out, const mdhip_array *scalar);
/* out[i] = start + i*step (numpy.py:125 arange) */
int mdhip_arange(const mdhip_array *out, double start, double step);

/* ======================= random fills (opt-in device RNG) ================= */
/* Counter-based stream (Philox4x32-10, csrc/md_rng.h): element i of a fill depends on (seed, offset, i) only, the CPU test
 * double produces the same words. kind 0: uniform [0,1) (f32/f64); 1: standard normal (f32/f64); 2: integers in [a, a+b)
 * (i32/i64; b = span <= 2^53); 3: binomial(n = a <= 256, p = b) (i32/i64). `out` C-contiguous. The caller advances `offset`
 * (in Philox blocks of four 32-bit words) past what a call consumed: ceil(elements * words per element / 4), words per element
 * = 1 / 2 (uniform f32 / f64), 2 / 4 (normal), 2 (integers), n (binomial).
 * Replaces, when asked to, rand / randn / randint / binomial of numpy.py:131-136 — whose default stays host NumPy. */
int mdhip_random_fill(int kind, uint64_t seed, uint64_t offset, double a, double b, const mdhip_array *out);
/* out (1-D int64, contiguous) = a uniformly random permutation of 0..n-1: indices sorted by a 64-bit key each (2 words per
 * element). numpy.py:135-136 permutation / shuffle gather with it. */
int mdhip_random_permutation(uint64_t seed, uint64_t offset, const mdhip_array *out);

/* ======================= reductions ======================================= */
/* Reduce x over the axes whose bit is set in axis_mask. `out` has x's ndim
 * with reduced axes of extent 1 (the caller drops them for keepdims=False).
 * SUM/PROD accumulate in out's dtype; ARG* write int64 flat positions along
 * the (single, or fully flattened) reduced extent.
 * (reference: numpy.py:20-23,43-46,57; reduce-to-shape definitions.py:157-183) */
int mdhip_reduce(int op, const mdhip_array *x, const mdhip_array *out, uint32_t axis_mask);
/* Variance / standard deviation along ONE axis in a single entry point. NumPy's np.std (numpy.py:57, called by definitions.py:209-221
 * in the forward pass and again in the vjp) is mean -> x - mean -> square -> sum -> divide -> sqrt: four reads and two writes of the
 * array when composed from the entry points above; this is one read (rows) or two (columns), same arithmetic up to the order of the
 * two sums. x: float32 / float64, C-contiguous, 16-B aligned; the reduced axis must be the last one (row length a multiple of 16 B)
 * or, for a 2-D view (n, inner), the first (inner >= 256 and a multiple of 16 B, n >= 64). out: C-contiguous, x's dtype, one element
 * per kept position; result = sum((x - mean)^2) / (n - ddof), square-rooted when take_sqrt != 0. Requires n - ddof > 0.
 * Any other form returns MDHIP_EVALUE and the caller composes the result from the other entry points (as NumPy does). */
int mdhip_var(const mdhip_array *x, const mdhip_array *out, int32_t axis, int64_t ddof, int take_sqrt);

/* ======================= matmul =========================================== */
/* C[b] = A[b] @ B[b]: A (batch.., M, K), B (batch.., K, N), C (batch.., M, N),
 * any strides (NN / NT / TN arrive as strided views: definitions.py:487-492).
 * Arrays are passed 3-D (batch, rows, cols); batch stride 0 broadcasts.
 * f32 and f64 run on MFMA; other dtypes take the generic kernel. */
int mdhip_matmul(const mdhip_array *a, const mdhip_array *b, const mdhip_array *c);
/* GEMM with the elementwise tail and the reduction of BASELINE's MLP forward in its epilogue (lazy mode,
 * minidiff_amd/ndarray.py recognises the pattern): for row-major float32 a (M x K), b (K x N), bias (N)
 *     sum_out  = sum(where(a @ b + bias > 0, a @ b + bias, 0))       (0-d float32)
 *     mask_out = (a @ b + bias > 0)                                  (M x N numpy.bool_, C-contiguous)
 * in ONE pass: the pre-activation is never written; the mask is what the backward pass needs of it
 * (reference call pattern: minidiff/ops/definitions.py:487-492 matmul, :424-427 add, :468-471 greater,
 * :555-559 where, :403-407 sum). Shapes that are not whole aligned tiles return MDHIP_EVALUE and the caller
 * runs the plain product followed by the fused tail. Deterministic (per-block partials summed in order). */
int mdhip_matmul_bias_relu_sum(const mdhip_array *a, const mdhip_array *b, const mdhip_array *bias,
                               const mdhip_array *mask_out, const mdhip_array *sum_out);

/* ======================= indexing (bit-exact) ============================= */
/* Generalised gather/scatter over an "indexed view":
 *   offset(p) = sum_d p[d]*src_strides[d]
 *             + sum_k wrap(idx_k[ sum_d p[d]*idx_strides[k][d] ], idx_extent[k]) * idx_mult[k]
 * for every position p of `shape[ndim]`. Covers a[key] with integer-array keys,
 * take_along_axis / put_along_axis, a[key] = v, and np.add.at.
 * (reference: numpy.py:73-75,105,108,124; definitions.py:186-189) */
typedef struct mdhip_index_plan {
  int32_t ndim;
  int32_t n_idx;
  int64_t shape[MDHIP_MAX_NDIM];
  int64_t src_strides[MDHIP_MAX_NDIM];
  const void *idx_ptr[MDHIP_MAX_NDIM];
  int32_t idx_dtype[MDHIP_MAX_NDIM];              /* MDHIP_I32 / MDHIP_I64 */
  int64_t idx_extent[MDHIP_MAX_NDIM];             /* axis length for wrap + bounds */
  int64_t idx_mult[MDHIP_MAX_NDIM];               /* source stride of that axis */
  int64_t idx_strides[MDHIP_MAX_NDIM][MDHIP_MAX_NDIM];
} mdhip_index_plan;
enum { MDHIP_SCATTER_SET = 0, MDHIP_SCATTER_ADD = 1 };
/* out[p] = src[offset(p)]; out is contiguous-or-strided with plan->shape. */
int mdhip_gather(const mdhip_index_plan *plan, const void *src, int dtype,
                 const mdhip_array *out);
/* dst[offset(p)] (=|+=) val[p]; duplicates accumulate in p order for ADD (the
 * np.add.at contract), last p wins for SET. Out-of-range -> MDHIP_EINDEX, before
 * anything is written (both calls read the verdict back: they SYNCHRONISE).
 * Between mdhip_graph_begin and _end neither call can synchronise, and a later
 * replay may meet other indices: there the kernels skip out-of-range positions
 * (gather) / write nothing at all (scatter), and MDHIP_EINDEX is returned ONCE by
 * the next mdhip_sync or mdhip_d2h after the replay that met them. A scatter that
 * needs host-driven rounds (element-wise duplicates, > 4096 positions, not whole
 * rows) fails the capture with MDHIP_ERUNTIME instead. */
int mdhip_scatter(const mdhip_index_plan *plan, void *dst, int dtype,
                  const mdhip_array *val, int mode);

/* Positions of the non-zero elements of a CONTIGUOUS array, ascending flat order
 * (np.flatnonzero; serves boolean-mask keys and np.argwhere, numpy.py:24,73-75).
 * Two calls: count first (SYNCHRONISES, the result size is data dependent), then
 * fill `out_flat` (int64[count]). */
int mdhip_nonzero_count(const mdhip_array *x, int64_t *count_out);
int mdhip_nonzero_fill(const mdhip_array *x, int64_t count, int64_t *out_flat);

/* ======================= fused expressions (opt-in lazy mode) ============== */
/* One pass over HBM for a whole elementwise expression, optionally ending in a
 * reduction (the "fused elementwise + reduce-to-shape backward" of the north
 * star). The caller (minidiff_amd/lazy.py) records chains of
 * backend calls instead of launching them and hands over a POSTFIX program that
 * a fixed interpreter kernel evaluates per element on a 4-deep register stack
 * (instructions are prefetched from LDS; most carry their leaf / constant operand
 * themselves): no runtime compilation. Values are held in `compute_dtype` (F32 or F64);
 * bool values travel as 0/1. Every operator applies the same functor as the
 * eager kernel (csrc/md_ops.h), so per-element results are identical. */
#define MDHIP_VM_MAX_INSTR 48
#define MDHIP_VM_MAX_LEAVES 8
#define MDHIP_VM_STACK 4
/* instruction kinds */
enum {
  MDHIP_VM_PUSH = 0, /* push rhs operand (leaf or const) */
  MDHIP_VM_UNARY,    /* s0 = f(s0), op = MDHIP_U_* */
  MDHIP_VM_BINARY,   /* op = MDHIP_B_*; operands: see below */
  MDHIP_VM_WHERE     /* s0 = s2 ? s1 : s0, pops 2 */
};
/* operand sources of a BINARY (and the rhs of PUSH) */
enum { MDHIP_VM_SRC_STACK = 0, MDHIP_VM_SRC_LEAF = 1, MDHIP_VM_SRC_CONST = 2 };
/* ctrl word: kind[0:2] op[3:7] lhs_src[8:9] lhs_leaf[10:12] rhs_src[13:14] rhs_leaf[15:17].
 * BINARY: value = op(lhs, rhs). Both STACK: lhs = s1, rhs = s0, pops one. One STACK:
 * it is s0, replaced in place (the common "x = op(x, leaf|const)" costs no stack traffic).
 * None STACK: the result is pushed. A CONST operand reads imm[pc] (so at most one per
 * instruction). */
#define MDHIP_VM_CTRL(kind, op, ls, ll, rs, rl) \
  ((uint32_t)(kind) | ((uint32_t)(op) << 3) | ((uint32_t)(ls) << 8) | ((uint32_t)(ll) << 10) | ((uint32_t)(rs) << 13) | ((uint32_t)(rl) << 15))
typedef struct mdhip_vm_program {
  int32_t n_instr, n_leaves, compute_dtype, _pad;
  uint32_t ctrl[MDHIP_VM_MAX_INSTR];
  double imm[MDHIP_VM_MAX_INSTR];
  mdhip_array leaves[MDHIP_VM_MAX_LEAVES]; /* each broadcast to out's shape (stride 0 on broadcast axes) */
} mdhip_vm_program;
/* out[...] = program(...)  (out dtype: compute_dtype, or BOOL for a 0/1 result) */
int mdhip_vm_eval(const mdhip_vm_program *prog, const mdhip_array *out);
/* out = reduce(program) with reduce_op in {SUM, PROD, MAX, MIN}. `shape` is the
 * program's shape (leaves are broadcast to it); supported forms: all axes reduced,
 * or a 2-D program reduced over axis 0 (reduce-to-shape of a row broadcast).
 * Anything else returns MDHIP_EVALUE and the caller materialises first. */
int mdhip_vm_reduce(const mdhip_vm_program *prog, int reduce_op, const mdhip_array *shape_like,
                    const mdhip_array *out, uint32_t axis_mask);

/* Large programs are specialised at run time: the same instruction sequence is
 * emitted as straight-line HIP source over the md_ops.h functors, compiled with
 * hiprtc for gfx950 and cached by signature; the interpreter remains the fallback
 * (small arrays, no libhiprtc, MDHIP_JIT=0). These two entry points are diagnostics:
 * compile-only check of a program (kind 0 = eval, 1 = full reduce, 2 = column
 * reduce (tiled), 3 = column reduce (sweep), 4 = eval + column reduce in one pass;
 * needs no device), and counters {kernels compiled, kernels launched}. Generated
 * kernels are named k_fused_<form>_<digest of the program signature>. */
int mdhip_vm_jit_probe(const mdhip_vm_program *prog, int kind, int reduce_op, int out_is_bool,
                       char *log, size_t log_capacity);
/* outs[k][...] = progs[k](...) for n programs of ONE shape (e.g. the gradients of one
 * backward sweep, which share their operands): with run-time specialisation available and
 * 2..4 programs whose merged operand tables fit one launch, every distinct leaf is read once
 * and all results are written in the same pass; otherwise exactly n mdhip_vm_eval calls. */
int mdhip_vm_eval_multi(const mdhip_vm_program *progs, const mdhip_array *outs, int n);
/* out_eval[r][c] = prog(r, c) AND out_red[c] = reduce over r (reduce_op in {SUM, PROD, MAX, MIN}) of a 2-D
 * program in ONE pass over its operands: the elementwise product of a backward step and its
 * reduce-to-shape (cfg4: g * mask -> W-gradient operand, and its column sum -> bias gradient;
 * reference call pattern minidiff/ops/definitions.py:555-559 then :157-183). Covered shapes:
 * rows of 1024/2048/4096/8192 elements, >= 512 rows, with run-time specialisation available;
 * anything else returns MDHIP_EVALUE and the caller runs the two passes separately. */
int mdhip_vm_eval_reduce_cols(const mdhip_vm_program *prog, int reduce_op, const mdhip_array *out_eval,
                              const mdhip_array *out_red);
int mdhip_vm_jit_probe_multi(const mdhip_vm_program *progs, int n, char *log, size_t log_capacity);
int mdhip_vm_jit_stats(int64_t stats[2]);

/* ======================= data-parallel (RCCL over xGMI) =================== */
/* One communicator per process (one process per GPU). uid is the 128-byte
 * ncclUniqueId produced on rank 0 and distributed by the launcher. */
#define MDHIP_UID_BYTES 128
/* Can this process reach RCCL at all (opens librccl; creates nothing)? Ranks agree on the answer BEFORE
 * ncclCommInitRank, which would wait for the missing ones for ever (minidiff_amd/dp.py RcclComm). */
int mdhip_comm_probe(void);
/* Number of ranks of the live communicator, as RCCL reports it (ncclCommCount). */
int mdhip_comm_count(int *nranks_out);
int mdhip_comm_get_unique_id(uint8_t uid[MDHIP_UID_BYTES]);
int mdhip_comm_init(int nranks, int rank, const uint8_t uid[MDHIP_UID_BYTES]);
int mdhip_comm_allreduce_sum(void *buf, size_t count, int dtype); /* in place, on the stream */
/* Overlapped form: the collective is issued on a second stream, after
 * everything enqueued so far on the compute stream, and runs beside the kernels enqueued
 * next (the rest of backward). mdhip_comm_wait makes the compute stream wait for it; buf must
 * not be released, read or written on the compute stream before that call. One in flight. */
int mdhip_comm_allreduce_sum_async(void *buf, size_t count, int dtype);
int mdhip_comm_wait(void);
int mdhip_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* MDHIP_H */
