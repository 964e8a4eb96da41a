// md_vm.h — the expression interpreter behind mdhip_vm_eval / mdhip_vm_reduce.
// Shared by the gfx950 kernels (fusion.hip) and the CPU test double.
//
// A program is a short postfix sequence over a 4-deep operand stack kept in named
// register arrays of W lanes (W = 4 on the vector path; static indices only, so
// nothing spills to scratch). To keep the per-instruction overhead below the HBM
// time of the data it touches:
//   * one interpreter step covers W lanes per thread (one 16-B vector group on the
//     device's vector path);
//   * most instructions are operand-fused (s0 = op(s0, leaf|const)), so a chain
//     costs one instruction per operator and no stack traffic;
//   * an instruction is 16 bytes {ctrl, pad, imm(double)}; Fetch supplies it — the
//     device fetches from LDS one instruction ahead (ds_read_b128 broadcast +
//     readfirstlane), the host double indexes the program directly.
// Control flow depends only on the program: uniform across a wavefront.
#pragma once
#include "md_common.h"

struct MdVmLeaf {
  const void *p;
  int64_t os;     // fast geometry: outer (row) stride, elements
  int32_t is;     // fast geometry: inner stride 0 / 1
  int32_t dtype;
};
struct MdVmInstr {
  uint32_t ctrl, pad;
  double imm;
};
struct MdVmDev {
  int32_t n_instr, n_leaves;
  MdVmLeaf leaf[MDHIP_VM_MAX_LEAVES];
  MdVmInstr code[MDHIP_VM_MAX_INSTR];
};
// generic geometry: (ndim, shape) shared, per-leaf strides
struct MdVmIter {
  int32_t ndim;
  int64_t total;
  int64_t shape[MDHIP_MAX_NDIM];
  int64_t strides[MDHIP_VM_MAX_LEAVES + 1][MDHIP_MAX_NDIM];  // last = out
};

#define MD_VM_KIND(c) ((c) & 7u)
#define MD_VM_OP(c) (((c) >> 3) & 31u)
#define MD_VM_LS(c) (((c) >> 8) & 3u)
#define MD_VM_LL(c) (((c) >> 10) & 7u)
#define MD_VM_RS(c) (((c) >> 13) & 3u)
#define MD_VM_RL(c) (((c) >> 15) & 7u)

// The opcode switch sits OUTSIDE the W-lane loop: one uniform branch per instruction.
#define MD_VM_U(code, expr) \
  case code:                \
    _Pragma("unroll") for (int j = 0; j < W; ++j) { const T x = s0[j]; s0[j] = (expr); } \
    break;
template <class T, int W> MD_HD void md_vm_unary(int op, T (&s0)[W]) {
  switch (op) {
    MD_VM_U(MDHIP_U_ABS, UAbs::apply(x))
    MD_VM_U(MDHIP_U_NEG, UNeg::apply(x))
    MD_VM_U(MDHIP_U_SIGN, USign::apply(x))
    MD_VM_U(MDHIP_U_CEIL, UCeil::apply(x))
    MD_VM_U(MDHIP_U_FLOOR, UFloor::apply(x))
    MD_VM_U(MDHIP_U_SIN, USin::apply(x))
    MD_VM_U(MDHIP_U_COS, UCos::apply(x))
    MD_VM_U(MDHIP_U_TAN, UTan::apply(x))
    MD_VM_U(MDHIP_U_SINH, USinh::apply(x))
    MD_VM_U(MDHIP_U_COSH, UCosh::apply(x))
    MD_VM_U(MDHIP_U_TANH, UTanh::apply(x))
    MD_VM_U(MDHIP_U_EXP, UExp::apply(x))
    MD_VM_U(MDHIP_U_LOG, ULog::apply(x))
    MD_VM_U(MDHIP_U_SQRT, USqrt::apply(x))
    MD_VM_U(MDHIP_U_LOGICAL_NOT, (T)(x == (T)0))
    MD_VM_U(MDHIP_U_ISNAN, (T)(x != x))
    default: break;  // COPY
  }
}
#undef MD_VM_U
// r[j] = op(a[j], b[j])
#define MD_VM_B(code, expr) \
  case code:                \
    _Pragma("unroll") for (int j = 0; j < W; ++j) { const T a = A[j], b = B[j]; r[j] = (expr); } \
    break;
template <class T, int W> MD_HD void md_vm_binary(int op, const T (&A)[W], const T (&B)[W], T (&r)[W]) {
  switch (op) {
    MD_VM_B(MDHIP_B_ADD, a + b)
    MD_VM_B(MDHIP_B_SUB, a - b)
    MD_VM_B(MDHIP_B_MUL, a * b)
    MD_VM_B(MDHIP_B_TRUE_DIV, a / b)
    MD_VM_B(MDHIP_B_FLOOR_DIV, BFloorDiv::apply(a, b))
    MD_VM_B(MDHIP_B_MOD, BMod::apply(a, b))
    MD_VM_B(MDHIP_B_POW, BPow::apply(a, b))
    MD_VM_B(MDHIP_B_MAXIMUM, BMaximum::apply(a, b))
    MD_VM_B(MDHIP_B_MINIMUM, BMinimum::apply(a, b))
    MD_VM_B(MDHIP_B_EQ, (T)(a == b))
    MD_VM_B(MDHIP_B_NE, (T)(a != b))
    MD_VM_B(MDHIP_B_LT, (T)(a < b))
    MD_VM_B(MDHIP_B_LE, (T)(a <= b))
    MD_VM_B(MDHIP_B_GT, (T)(a > b))
    MD_VM_B(MDHIP_B_GE, (T)(a >= b))
    MD_VM_B(MDHIP_B_LAND, (T)((a != (T)0) && (b != (T)0)))
    MD_VM_B(MDHIP_B_LOR, (T)((a != (T)0) || (b != (T)0)))
    MD_VM_B(MDHIP_B_LXOR, (T)((a != (T)0) != (b != (T)0)))
    default:
#pragma unroll
      for (int j = 0; j < W; ++j) r[j] = A[j];
  }
}
#undef MD_VM_B

// Loader concept: void operator()(int leaf, T (&dst)[W]) — loads W lanes of leaf `leaf`.
template <class T, int W, class Loader>
MD_HD void md_vm_operand(uint32_t src, uint32_t leaf, double imm, Loader &load, T (&d)[W]) {
  if (src == MDHIP_VM_SRC_CONST) {
    const T c = (T)imm;
#pragma unroll
    for (int j = 0; j < W; ++j) d[j] = c;
  } else {
    load((int)leaf, d);
  }
}

// Fetch concept: void operator()(int pc, uint32_t &ctrl, double &imm); prefetch(int pc).
template <class T, int W, class Fetch, class Loader>
MD_HD void md_vm_run(const int32_t n_instr, Fetch &fetch, Loader &load, T (&s0)[W]) {
  T s1[W], s2[W], s3[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { s0[j] = (T)0; s1[j] = (T)0; s2[j] = (T)0; s3[j] = (T)0; }
  fetch.prefetch(0);
  for (int pc = 0; pc < n_instr; ++pc) {
    uint32_t c;
    double imm;
    fetch(pc, c, imm);
    fetch.prefetch(pc + 1);
    const uint32_t k = MD_VM_KIND(c);
    if (k == MDHIP_VM_BINARY) {
      const uint32_t ls = MD_VM_LS(c), rs = MD_VM_RS(c), op = MD_VM_OP(c);
      if (ls == MDHIP_VM_SRC_STACK && rs == MDHIP_VM_SRC_STACK) {
        T t[W];
        md_vm_binary<T, W>(op, s1, s0, t);
#pragma unroll
        for (int j = 0; j < W; ++j) { s0[j] = t[j]; s1[j] = s2[j]; s2[j] = s3[j]; }
      } else if (ls == MDHIP_VM_SRC_STACK) {
        T o[W], t[W];
        md_vm_operand<T, W>(rs, MD_VM_RL(c), imm, load, o);
        md_vm_binary<T, W>(op, s0, o, t);
#pragma unroll
        for (int j = 0; j < W; ++j) s0[j] = t[j];
      } else if (rs == MDHIP_VM_SRC_STACK) {
        T o[W], t[W];
        md_vm_operand<T, W>(ls, MD_VM_LL(c), imm, load, o);
        md_vm_binary<T, W>(op, o, s0, t);
#pragma unroll
        for (int j = 0; j < W; ++j) s0[j] = t[j];
      } else {
        T o1[W], o2[W];
        md_vm_operand<T, W>(ls, MD_VM_LL(c), imm, load, o1);
        md_vm_operand<T, W>(rs, MD_VM_RL(c), imm, load, o2);
#pragma unroll
        for (int j = 0; j < W; ++j) { s3[j] = s2[j]; s2[j] = s1[j]; s1[j] = s0[j]; }
        md_vm_binary<T, W>(op, o1, o2, s0);
      }
    } else if (k == MDHIP_VM_UNARY) {
      md_vm_unary<T, W>(MD_VM_OP(c), s0);
    } else if (k == MDHIP_VM_PUSH) {
#pragma unroll
      for (int j = 0; j < W; ++j) { s3[j] = s2[j]; s2[j] = s1[j]; s1[j] = s0[j]; }
      md_vm_operand<T, W>(MD_VM_RS(c), MD_VM_RL(c), imm, load, s0);
    } else {  // WHERE
#pragma unroll
      for (int j = 0; j < W; ++j) { s0[j] = (s2[j] != (T)0) ? s1[j] : s0[j]; s1[j] = s3[j]; }
    }
  }
}

// direct fetch from a program in addressable memory (host double; device fallback)
struct MdVmFetchDirect {
  const uint32_t *ctrl;
  const double *imm;
  MD_HD void prefetch(int) {}
  MD_HD void operator()(int pc, uint32_t &c, double &i) const { c = ctrl[pc]; i = imm[pc]; }
};

// ---- validation + geometry, shared by both builds ---------------------------------
static inline int md_vm_check(const mdhip_vm_program *pr) {
  if (!pr) return md_fail(MDHIP_EVALUE, "vm: null program");
  if (pr->n_instr < 1 || pr->n_instr > MDHIP_VM_MAX_INSTR) return md_fail(MDHIP_EVALUE, "vm: %d instructions (max %d)", pr->n_instr, MDHIP_VM_MAX_INSTR);
  if (pr->n_leaves < 0 || pr->n_leaves > MDHIP_VM_MAX_LEAVES) return md_fail(MDHIP_EVALUE, "vm: %d leaves (max %d)", pr->n_leaves, MDHIP_VM_MAX_LEAVES);
  if (pr->compute_dtype != MDHIP_F32 && pr->compute_dtype != MDHIP_F64) return md_fail(MDHIP_ETYPE, "vm: compute dtype must be float32 or float64");
  int depth = 0;
  for (int pc = 0; pc < pr->n_instr; ++pc) {
    const uint32_t c = pr->ctrl[pc];
    const uint32_t k = MD_VM_KIND(c), op = MD_VM_OP(c), ls = MD_VM_LS(c), rs = MD_VM_RS(c);
    if (ls == MDHIP_VM_SRC_LEAF && (int)MD_VM_LL(c) >= pr->n_leaves) return md_fail(MDHIP_EVALUE, "vm: leaf index out of range at %d", pc);
    if (rs == MDHIP_VM_SRC_LEAF && (int)MD_VM_RL(c) >= pr->n_leaves) return md_fail(MDHIP_EVALUE, "vm: leaf index out of range at %d", pc);
    if (ls > MDHIP_VM_SRC_CONST || rs > MDHIP_VM_SRC_CONST) return md_fail(MDHIP_EVALUE, "vm: bad operand source at %d", pc);
    switch (k) {
      case MDHIP_VM_PUSH:
        if (rs == MDHIP_VM_SRC_STACK) return md_fail(MDHIP_EVALUE, "vm: PUSH needs a leaf or const at %d", pc);
        ++depth;
        break;
      case MDHIP_VM_UNARY:
        if (depth < 1 || op >= MDHIP_U_COUNT || op == MDHIP_U_INVERT) return md_fail(MDHIP_EVALUE, "vm: bad unary at %d", pc);
        break;
      case MDHIP_VM_BINARY: {
        if (op >= MDHIP_B_COUNT) return md_fail(MDHIP_EVALUE, "vm: bad binary op at %d", pc);
        if (ls == MDHIP_VM_SRC_CONST && rs == MDHIP_VM_SRC_CONST) return md_fail(MDHIP_EVALUE, "vm: two const operands at %d", pc);
        const int need = (ls == MDHIP_VM_SRC_STACK) + (rs == MDHIP_VM_SRC_STACK);
        if (depth < need) return md_fail(MDHIP_EVALUE, "vm: stack underflow at %d", pc);
        depth += 1 - need;
      } break;
      case MDHIP_VM_WHERE:
        if (depth < 3) return md_fail(MDHIP_EVALUE, "vm: where needs 3 operands at %d", pc);
        depth -= 2;
        break;
      default: return md_fail(MDHIP_EVALUE, "vm: unknown instruction kind %u", k);
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
