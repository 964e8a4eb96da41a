// md_vm.h — the postfix expression interpreter behind mdhip_vm_eval / mdhip_vm_reduce.
// Shared by the gfx950 kernels (fusion.hip) and the CPU test double. The stack is
// four named register arrays of W lanes (W = 4 on the vector path), shifted on
// push/pop with static indices only, so nothing is spilled to scratch; control
// flow depends only on the program, i.e. it is uniform across a wavefront.
#pragma once
#include "md_common.h"

struct MdVmLeaf {
  const void *p;
  int64_t os;     // fast geometry: outer (row) stride, elements
  int32_t is;     // fast geometry: inner stride 0 / 1
  int32_t dtype;
};
struct MdVmDev {
  int32_t n_instr, n_leaves;
  uint8_t kind[MDHIP_VM_MAX_INSTR];
  uint8_t arg[MDHIP_VM_MAX_INSTR];
  double consts[MDHIP_VM_MAX_CONSTS];
  MdVmLeaf leaf[MDHIP_VM_MAX_LEAVES];
};
// generic geometry: (ndim, shape) shared, per-leaf strides
struct MdVmIter {
  int32_t ndim;
  int64_t total;
  int64_t shape[MDHIP_MAX_NDIM];
  int64_t strides[MDHIP_VM_MAX_LEAVES + 1][MDHIP_MAX_NDIM];  // last = out
};

template <class T> MD_HD T md_vm_unary(int op, T x) {
  switch (op) {
    case MDHIP_U_COPY: return x;
    case MDHIP_U_ABS: return UAbs::apply(x);
    case MDHIP_U_NEG: return UNeg::apply(x);
    case MDHIP_U_SIGN: return USign::apply(x);
    case MDHIP_U_CEIL: return UCeil::apply(x);
    case MDHIP_U_FLOOR: return UFloor::apply(x);
    case MDHIP_U_SIN: return USin::apply(x);
    case MDHIP_U_COS: return UCos::apply(x);
    case MDHIP_U_TAN: return UTan::apply(x);
    case MDHIP_U_SINH: return USinh::apply(x);
    case MDHIP_U_COSH: return UCosh::apply(x);
    case MDHIP_U_TANH: return UTanh::apply(x);
    case MDHIP_U_EXP: return UExp::apply(x);
    case MDHIP_U_LOG: return ULog::apply(x);
    case MDHIP_U_SQRT: return USqrt::apply(x);
    case MDHIP_U_LOGICAL_NOT: return (T)(x == (T)0);
    case MDHIP_U_ISNAN: return (T)(x != x);
  }
  return x;
}
template <class T> MD_HD T md_vm_binary(int op, T a, T b) {
  switch (op) {
    case MDHIP_B_ADD: return a + b;
    case MDHIP_B_SUB: return a - b;
    case MDHIP_B_MUL: return a * b;
    case MDHIP_B_TRUE_DIV: return a / b;
    case MDHIP_B_FLOOR_DIV: return BFloorDiv::apply(a, b);
    case MDHIP_B_MOD: return BMod::apply(a, b);
    case MDHIP_B_POW: return BPow::apply(a, b);
    case MDHIP_B_MAXIMUM: return BMaximum::apply(a, b);
    case MDHIP_B_MINIMUM: return BMinimum::apply(a, b);
    case MDHIP_B_EQ: return (T)(a == b);
    case MDHIP_B_NE: return (T)(a != b);
    case MDHIP_B_LT: return (T)(a < b);
    case MDHIP_B_LE: return (T)(a <= b);
    case MDHIP_B_GT: return (T)(a > b);
    case MDHIP_B_GE: return (T)(a >= b);
    case MDHIP_B_LAND: return (T)((a != (T)0) && (b != (T)0));
    case MDHIP_B_LOR: return (T)((a != (T)0) || (b != (T)0));
    case MDHIP_B_LXOR: return (T)((a != (T)0) != (b != (T)0));
  }
  return a;
}

// Loader concept: void operator()(int leaf, T (&dst)[W]) — fills W lanes of leaf `leaf`.
template <class T, int W, class Loader>
MD_HD void md_vm_run(const int32_t n_instr, const uint8_t *kind, const uint8_t *arg, const double *consts, Loader &load,
                     T (&s0)[W]) {
  T s1[W], s2[W], s3[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { s0[j] = (T)0; s1[j] = (T)0; s2[j] = (T)0; s3[j] = (T)0; }
  for (int pc = 0; pc < n_instr; ++pc) {
    const int k = kind[pc], a = arg[pc];
    if (k == MDHIP_VM_PUSH_LEAF || k == MDHIP_VM_PUSH_CONST) {
#pragma unroll
      for (int j = 0; j < W; ++j) { s3[j] = s2[j]; s2[j] = s1[j]; s1[j] = s0[j]; }
      if (k == MDHIP_VM_PUSH_LEAF) {
        load(a, s0);
      } else {
        const T c = (T)consts[a];
#pragma unroll
        for (int j = 0; j < W; ++j) s0[j] = c;
      }
    } else if (k == MDHIP_VM_UNARY) {
#pragma unroll
      for (int j = 0; j < W; ++j) s0[j] = md_vm_unary<T>(a, s0[j]);
    } else if (k == MDHIP_VM_BINARY) {
#pragma unroll
      for (int j = 0; j < W; ++j) { s0[j] = md_vm_binary<T>(a, s1[j], s0[j]); s1[j] = s2[j]; s2[j] = s3[j]; }
    } else {  // WHERE
#pragma unroll
      for (int j = 0; j < W; ++j) { s0[j] = (s2[j] != (T)0) ? s1[j] : s0[j]; s1[j] = s3[j]; }
    }
  }
}

// ---- validation + geometry, shared by both builds ---------------------------------
static inline int md_vm_check(const mdhip_vm_program *pr) {
  if (!pr) return md_fail(MDHIP_EVALUE, "vm: null program");
  if (pr->n_instr < 1 || pr->n_instr > MDHIP_VM_MAX_INSTR) return md_fail(MDHIP_EVALUE, "vm: %d instructions (max %d)", pr->n_instr, MDHIP_VM_MAX_INSTR);
  if (pr->n_leaves < 0 || pr->n_leaves > MDHIP_VM_MAX_LEAVES) return md_fail(MDHIP_EVALUE, "vm: %d leaves (max %d)", pr->n_leaves, MDHIP_VM_MAX_LEAVES);
  if (pr->n_consts < 0 || pr->n_consts > MDHIP_VM_MAX_CONSTS) return md_fail(MDHIP_EVALUE, "vm: %d consts (max %d)", pr->n_consts, MDHIP_VM_MAX_CONSTS);
  if (pr->compute_dtype != MDHIP_F32 && pr->compute_dtype != MDHIP_F64) return md_fail(MDHIP_ETYPE, "vm: compute dtype must be float32 or float64");
  int depth = 0;
  for (int pc = 0; pc < pr->n_instr; ++pc) {
    const int k = pr->kind[pc], a = pr->arg[pc];
    switch (k) {
      case MDHIP_VM_PUSH_LEAF: if (a >= pr->n_leaves) return md_fail(MDHIP_EVALUE, "vm: leaf index %d out of range", a); ++depth; break;
      case MDHIP_VM_PUSH_CONST: if (a >= pr->n_consts) return md_fail(MDHIP_EVALUE, "vm: const index %d out of range", a); ++depth; break;
      case MDHIP_VM_UNARY: if (depth < 1 || a >= MDHIP_U_COUNT || a == MDHIP_U_INVERT) return md_fail(MDHIP_EVALUE, "vm: bad unary at %d", pc); break;
      case MDHIP_VM_BINARY: if (depth < 2 || a >= MDHIP_B_COUNT) return md_fail(MDHIP_EVALUE, "vm: bad binary at %d", pc); --depth; break;
      case MDHIP_VM_WHERE: if (depth < 3) return md_fail(MDHIP_EVALUE, "vm: where needs 3 operands at %d", pc); depth -= 2; break;
      default: return md_fail(MDHIP_EVALUE, "vm: unknown instruction kind %d", k);
    }
    if (depth > MDHIP_VM_STACK) return md_fail(MDHIP_EVALUE, "vm: stack depth %d exceeds %d", depth, MDHIP_VM_STACK);
  }
  if (depth != 1) return md_fail(MDHIP_EVALUE, "vm: program leaves %d values on the stack", depth);
  for (int l = 0; l < pr->n_leaves; ++l) MD_TRY(md_check_array(&pr->leaves[l], "vm leaf"));
  return MDHIP_OK;
}

// Collapsed iteration space over (leaves..., out); same merge rule as md_build_iter.
static inline int md_vm_build_iter(MdVmIter *it, const mdhip_vm_program *pr, const mdhip_array *shape_from, const mdhip_array *out) {
  const int nd = shape_from->ndim;
  const int nl = pr->n_leaves;
  if (nd < 0 || nd > MDHIP_MAX_NDIM) return md_fail(MDHIP_EVALUE, "vm: ndim out of range");
  for (int l = 0; l < nl; ++l) {
    const mdhip_array *a = &pr->leaves[l];
    if (a->is_scalar) return md_fail(MDHIP_EVALUE, "vm: scalar leaves must be consts");
    if (a->ndim != nd) return md_fail(MDHIP_EVALUE, "vm: leaf %d ndim mismatch", l);
    for (int d = 0; d < nd; ++d)
      if (a->shape[d] != shape_from->shape[d]) return md_fail(MDHIP_EVALUE, "vm: leaf %d shape mismatch on axis %d", l, d);
  }
  int64_t shp[MDHIP_MAX_NDIM];
  int64_t str[MDHIP_VM_MAX_LEAVES + 1][MDHIP_MAX_NDIM];
  it->total = 1;
  int m = 0;
  for (int d = 0; d < nd; ++d) {
    const int64_t e = shape_from->shape[d];
    it->total *= e;
    if (e == 1) continue;
    shp[m] = e;
    for (int l = 0; l < nl; ++l) str[l][m] = pr->leaves[l].strides[d];
    str[nl][m] = out ? out->strides[d] : 0;
    ++m;
  }
  int w = 0;
  for (int d = 0; d < m; ++d) {
    if (w > 0) {
      bool ok = true;
      for (int l = 0; l <= nl; ++l)
        if (str[l][w - 1] != str[l][d] * shp[d]) { ok = false; break; }
      if (ok) {
        shp[w - 1] *= shp[d];
        for (int l = 0; l <= nl; ++l) str[l][w - 1] = str[l][d];
        continue;
      }
    }
    shp[w] = shp[d];
    for (int l = 0; l <= nl; ++l) str[l][w] = str[l][d];
    ++w;
  }
  it->ndim = w;
  for (int d = 0; d < MDHIP_MAX_NDIM; ++d) {
    it->shape[d] = d < w ? shp[d] : 1;
    for (int l = 0; l <= MDHIP_VM_MAX_LEAVES; ++l) it->strides[l][d] = 0;
    if (d < w) {
      for (int l = 0; l < nl; ++l) it->strides[l][d] = str[l][d];
      it->strides[MDHIP_VM_MAX_LEAVES][d] = str[nl][d];
    }
  }
  return MDHIP_OK;
}
