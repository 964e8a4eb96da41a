/* _fastpath — the host side of one eager backend call, in C.
 *
 * An eager call of the backend table (`add(a, b)`, `multiply(a, 2.0)`, `sin(x)` .. — reference: minidiff/backend/numpy.py:19-92
 * as called from minidiff/ops/definitions.py:266-536) used to cost ~7 us of Python before the C-ABI was reached: operand
 * classification, NumPy loop resolution, broadcasting, a ctypes descriptor per operand, a Python owner object per result block,
 * a ctypes call. This CPython extension does that work for the common cases in C and calls the SAME C-ABI entry points
 * (mdhip_alloc / mdhip_unary / mdhip_binary / mdhip_reduce, include/mdhip.h) through function pointers taken from whichever
 * library the process is bound to (`bind`), so it is a shortcut through the host code, not a second compute path:
 *
 *   Buffer      owner of one allocator block (what ndarray._Buffer was), freed by mdhip_free when the last view dies
 *   ArrayBase   the fields of ndarray.DeviceArray as a C struct (DeviceArray subclasses it and adds the methods)
 *   FastOp      a callable per table entry: tries the C route, otherwise calls the Python implementation it wraps
 *
 * The C route takes: float32 / float64 device arrays that own valid memory (no pending lazy expression, no deferred fill),
 * optionally one weak Python scalar (int / float) or a bool array next to a float one, any strides, NumPy broadcasting, eager mode —
 * for the elementwise entries, `where` (bool condition), 2-D `matmul`, and sum / prod / max / min of arrays below 2^18 elements. Everything else — other dtypes,
 * NumPy scalars, keyword arguments, lazy mode, storage-only dtypes, shape errors — goes to the Python implementation, which
 * also owns every error message: when the C route cannot serve a call, or the C-ABI returns a non-zero status, the call is
 * simply repeated there.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <structmember.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

#include "mdhip.h"

typedef int (*alloc_fn)(size_t, void **);
typedef int (*free_fn)(void *);
typedef int (*unary_fn)(int, const mdhip_array *, const mdhip_array *);
typedef int (*binary_fn)(int, const mdhip_array *, const mdhip_array *, const mdhip_array *, int);
typedef int (*reduce_fn)(int, const mdhip_array *, const mdhip_array *, uint32_t);
typedef int (*matmul_fn)(const mdhip_array *, const mdhip_array *, const mdhip_array *);
typedef int (*where_fn)(const mdhip_array *, const mdhip_array *, const mdhip_array *, const mdhip_array *);

static struct {
  alloc_fn alloc;
  free_fn release;
  unary_fn unary;
  binary_fn binary;
  reduce_fn reduce;
  matmul_fn matmul;
  where_fn where;
  PyTypeObject *array_type;              /* ndarray.DeviceArray */
  PyObject *dtypes[MDHIP_NUM_DTYPES];    /* np.dtype by compute dtype code */
  PyObject *dtype_code;                  /* callable: np.dtype -> code (raises TypeError for an unsupported dtype) */
  PyObject *raise_status;                /* callable: C-ABI status -> raises the mapped exception with mdhip_last_error() */
  PyObject *loader;                      /* callable: binds the process-wide library (ndarray._lib) */
  int ops_enabled;                       /* FastOp tries the C route */
  int lazy;                              /* ndarray._LAZY mirror */
  unsigned long long served, passed;     /* FastOp calls answered in C / handed to Python */
} G;

static const int ITEMSIZE[MDHIP_NUM_DTYPES] = {1, 4, 8, 4, 8};

/* =========================================================================== Buffer */
typedef struct {
  PyObject_HEAD
  unsigned long long ptr;
  Py_ssize_t nbytes;
  free_fn release;
  PyObject *deps; /* {id: weakref} of pending (lazy) arrays that read this block */
  PyObject *task; /* deferred fill of this block (lazy mode) */
  PyObject *weakrefs;
} BufferObject;

static PyTypeObject Buffer_Type;

static int ensure_bound(void) {
  if (G.alloc) return 0;
  if (!G.loader) {
    PyErr_SetString(PyExc_RuntimeError, "_fastpath: no library bound");
    return -1;
  }
  PyObject *r = PyObject_CallNoArgs(G.loader);
  if (!r) return -1;
  Py_DECREF(r);
  if (!G.alloc) {
    PyErr_SetString(PyExc_RuntimeError, "_fastpath: the library loader did not bind the C-ABI");
    return -1;
  }
  return 0;
}

/* status != 0 -> the mapped Python exception (always returns NULL) */
static PyObject *raise_status(int st) {
  if (G.raise_status) {
    PyObject *r = PyObject_CallFunction(G.raise_status, "i", st);
    Py_XDECREF(r);
    if (PyErr_Occurred()) return NULL;
  }
  PyErr_Format(PyExc_RuntimeError, "libmdhip call failed with status %d", st);
  return NULL;
}

/* quiet != 0: a failed allocation returns NULL WITHOUT an exception (the caller repeats the call in Python) */
static BufferObject *buffer_create(Py_ssize_t nbytes, int quiet) {
  void *p = NULL;
  int st = G.alloc((size_t)(nbytes > 0 ? nbytes : 1), &p);
  if (st) {
    if (!quiet) raise_status(st);
    return NULL;
  }
  BufferObject *b = PyObject_GC_New(BufferObject, &Buffer_Type);
  if (!b) {
    G.release(p);
    return NULL;
  }
  b->ptr = (unsigned long long)(uintptr_t)p;
  b->nbytes = nbytes;
  b->release = G.release;
  b->deps = b->task = b->weakrefs = NULL;
  PyObject_GC_Track((PyObject *)b);
  return b;
}

static PyObject *Buffer_new(PyTypeObject *type, PyObject *args, PyObject *kw) {
  Py_ssize_t nbytes;
  if (kw && PyDict_GET_SIZE(kw)) {
    PyErr_SetString(PyExc_TypeError, "Buffer() takes no keyword arguments");
    return NULL;
  }
  if (!PyArg_ParseTuple(args, "n", &nbytes)) return NULL;
  if (type != &Buffer_Type) {
    PyErr_SetString(PyExc_TypeError, "Buffer cannot be subclassed");
    return NULL;
  }
  if (ensure_bound() < 0) return NULL;
  return (PyObject *)buffer_create(nbytes, 0);
}

static int Buffer_traverse(BufferObject *b, visitproc visit, void *arg) {
  Py_VISIT(b->deps);
  Py_VISIT(b->task);
  return 0;
}

static int Buffer_clear(BufferObject *b) {
  Py_CLEAR(b->deps);
  Py_CLEAR(b->task);
  return 0;
}

static void Buffer_dealloc(BufferObject *b) {
  PyObject_GC_UnTrack(b);
  if (b->weakrefs) PyObject_ClearWeakRefs((PyObject *)b);
  Buffer_clear(b);
  if (b->ptr && b->release) {
    b->release((void *)(uintptr_t)b->ptr);
    b->ptr = 0;
  }
  PyObject_GC_Del(b);
}

static PyMemberDef Buffer_members[] = {
    {"ptr", T_ULONGLONG, offsetof(BufferObject, ptr), READONLY, "device address of the block"},
    {"nbytes", T_PYSSIZET, offsetof(BufferObject, nbytes), READONLY, "requested size"},
    {"deps", T_OBJECT, offsetof(BufferObject, deps), 0, "pending readers of this block (lazy mode)"},
    {"task", T_OBJECT, offsetof(BufferObject, task), 0, "deferred fill of this block (lazy mode)"},
    {NULL}};

static PyTypeObject Buffer_Type = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "minidiff_amd._fastpath.Buffer",
    .tp_basicsize = sizeof(BufferObject),
    .tp_dealloc = (destructor)Buffer_dealloc,
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC,
    .tp_doc = "Owner of one allocator block; freed when the last view drops it.",
    .tp_traverse = (traverseproc)Buffer_traverse,
    .tp_clear = (inquiry)Buffer_clear,
    .tp_weaklistoffset = offsetof(BufferObject, weakrefs),
    .tp_members = Buffer_members,
    .tp_new = Buffer_new,
};

/* =========================================================================== ArrayBase */
typedef struct {
  PyObject_HEAD
  PyObject *buf;      /* Buffer | None (pending) */
  Py_ssize_t offset;  /* elements */
  PyObject *shape;    /* tuple of int */
  PyObject *strides;  /* tuple of int, elements */
  PyObject *dtype;    /* np.dtype */
  int code;
  PyObject *expr, *cdesc, *tasks, *dependents;
  PyObject *weakrefs;
} ArrayObject;

static PyTypeObject ArrayBase_Type;

static int Array_init(ArrayObject *a, PyObject *args, PyObject *kw) {
  PyObject *buf, *shape, *strides, *dtype;
  Py_ssize_t offset;
  int code = -1;
  if (kw && PyDict_GET_SIZE(kw)) {
    static char *names[] = {"buf", "offset", "shape", "strides", "dtype", "code", NULL};
    if (!PyArg_ParseTupleAndKeywords(args, kw, "OnOOO|i", names, &buf, &offset, &shape, &strides, &dtype, &code)) return -1;
  } else if (!PyArg_ParseTuple(args, "OnOOO|i", &buf, &offset, &shape, &strides, &dtype, &code)) {
    return -1;
  }
  if (!PyTuple_CheckExact(shape) || !PyTuple_CheckExact(strides)) {
    PyErr_SetString(PyExc_TypeError, "DeviceArray: shape and strides must be tuples");
    return -1;
  }
  if (code < 0) {
    if (!G.dtype_code) {
      PyErr_SetString(PyExc_RuntimeError, "_fastpath: not configured");
      return -1;
    }
    PyObject *c = PyObject_CallOneArg(G.dtype_code, dtype);
    if (!c) return -1;
    code = (int)PyLong_AsLong(c);
    Py_DECREF(c);
    if (code == -1 && PyErr_Occurred()) return -1;
  }
  Py_INCREF(buf);
  Py_XSETREF(a->buf, buf);
  a->offset = offset;
  Py_INCREF(shape);
  Py_XSETREF(a->shape, shape);
  Py_INCREF(strides);
  Py_XSETREF(a->strides, strides);
  Py_INCREF(dtype);
  Py_XSETREF(a->dtype, dtype);
  a->code = code;
  Py_CLEAR(a->expr);
  Py_CLEAR(a->cdesc);
  Py_CLEAR(a->tasks);
  Py_CLEAR(a->dependents);
  return 0;
}

static int Array_traverse(ArrayObject *a, visitproc visit, void *arg) {
  Py_VISIT(a->buf);
  Py_VISIT(a->shape);
  Py_VISIT(a->strides);
  Py_VISIT(a->dtype);
  Py_VISIT(a->expr);
  Py_VISIT(a->cdesc);
  Py_VISIT(a->tasks);
  Py_VISIT(a->dependents);
  return 0;
}

static int Array_clear(ArrayObject *a) {
  Py_CLEAR(a->buf);
  Py_CLEAR(a->shape);
  Py_CLEAR(a->strides);
  Py_CLEAR(a->dtype);
  Py_CLEAR(a->expr);
  Py_CLEAR(a->cdesc);
  Py_CLEAR(a->tasks);
  Py_CLEAR(a->dependents);
  return 0;
}

static void Array_dealloc(ArrayObject *a) {
  PyObject_GC_UnTrack(a);
  if (a->weakrefs) PyObject_ClearWeakRefs((PyObject *)a);
  Array_clear(a);
  Py_TYPE(a)->tp_free((PyObject *)a);
}

static PyMemberDef Array_members[] = {
    {"_buf", T_OBJECT, offsetof(ArrayObject, buf), 0, "allocator block (None while a lazy expression is pending)"},
    {"_offset", T_PYSSIZET, offsetof(ArrayObject, offset), 0, "offset of element 0 in the block, in elements"},
    {"shape", T_OBJECT, offsetof(ArrayObject, shape), 0, NULL},
    {"_strides", T_OBJECT, offsetof(ArrayObject, strides), 0, "element strides"},
    {"dtype", T_OBJECT, offsetof(ArrayObject, dtype), 0, NULL},
    {"_code", T_INT, offsetof(ArrayObject, code), 0, "MDHIP_* dtype code"},
    {"_expr", T_OBJECT, offsetof(ArrayObject, expr), 0, "pending expression (lazy mode)"},
    {"_cdesc", T_OBJECT, offsetof(ArrayObject, cdesc), 0, "own-shape ctypes descriptor, built once"},
    {"_tasks", T_OBJECT, offsetof(ArrayObject, tasks), 0, NULL},
    {"_dependents", T_OBJECT, offsetof(ArrayObject, dependents), 0, NULL},
    {NULL}};

static PyTypeObject ArrayBase_Type = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "minidiff_amd._fastpath.ArrayBase",
    .tp_basicsize = sizeof(ArrayObject),
    .tp_dealloc = (destructor)Array_dealloc,
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC | Py_TPFLAGS_BASETYPE,
    .tp_doc = "Fields of a DeviceArray: (buf, offset, shape, strides, dtype, code=-1).",
    .tp_traverse = (traverseproc)Array_traverse,
    .tp_clear = (inquiry)Array_clear,
    .tp_weaklistoffset = offsetof(ArrayObject, weakrefs),
    .tp_members = Array_members,
    .tp_init = (initproc)Array_init,
    .tp_new = PyType_GenericNew,
};

/* =========================================================================== the C route */
typedef struct {
  int is_array;
  int code;             /* array: its dtype code; scalar: the code it travels under (I64 / F64) */
  int ndim;
  int64_t shape[MDHIP_MAX_NDIM], strides[MDHIP_MAX_NDIM];
  ArrayObject *arr;
  int64_t si;
  double sf;
} Operand;

static inline int tuple_to_i64(PyObject *t, int64_t *out, int n) {
  for (int i = 0; i < n; i++) {
    PyObject *v = PyTuple_GET_ITEM(t, i);
    if (!PyLong_CheckExact(v)) return -1;
    long long x = PyLong_AsLongLong(v);
    if (x == -1 && PyErr_Occurred()) {
      PyErr_Clear();
      return -1;
    }
    out[i] = (int64_t)x;
  }
  return 0;
}

/* 1: usable operand; 0: not for the C route (no exception pending) */
static int parse_operand_ex(PyObject *x, Operand *o, int allow_bool) {
  if (Py_TYPE(x) == G.array_type) {
    ArrayObject *a = (ArrayObject *)x;
    if (a->code != MDHIP_F32 && a->code != MDHIP_F64 && !(allow_bool && a->code == MDHIP_BOOL)) return 0;
    if (!a->buf || Py_TYPE(a->buf) != &Buffer_Type) return 0;           /* pending, or a block of another owner type */
    BufferObject *b = (BufferObject *)a->buf;
    if ((b->task && b->task != Py_None) || !b->ptr) return 0;           /* deferred fill still owed */
    if (a->expr && a->expr != Py_None) return 0;
    if (!a->shape || !a->strides) return 0;
    Py_ssize_t nd = PyTuple_GET_SIZE(a->shape);
    if (nd > MDHIP_MAX_NDIM || PyTuple_GET_SIZE(a->strides) != nd) return 0;
    if (tuple_to_i64(a->shape, o->shape, (int)nd) < 0 || tuple_to_i64(a->strides, o->strides, (int)nd) < 0) return 0;
    o->is_array = 1;
    o->code = a->code;
    o->ndim = (int)nd;
    o->arr = a;
    return 1;
  }
  if (PyFloat_CheckExact(x)) {
    o->is_array = 0;
    o->code = MDHIP_F64;
    o->sf = PyFloat_AS_DOUBLE(x);
    o->si = 0;
    o->ndim = 0;
    o->arr = NULL;
    return 1;
  }
  if (PyLong_CheckExact(x)) { /* (bool is a subclass: not exact) */
    int overflow = 0;
    long long v = PyLong_AsLongLongAndOverflow(x, &overflow);
    if (overflow) return 0;
    if (v == -1 && PyErr_Occurred()) {
      PyErr_Clear();
      return 0;
    }
    o->is_array = 0;
    o->code = MDHIP_I64;
    o->si = (int64_t)v;
    o->sf = 0.0;
    o->ndim = 0;
    o->arr = NULL;
    return 1;
  }
  return 0;
}

static inline int parse_operand(PyObject *x, Operand *o) { return parse_operand_ex(x, o, 0); }

static inline void array_desc(const Operand *o, mdhip_array *d) {
  const ArrayObject *a = o->arr;
  d->data = (void *)(uintptr_t)(((BufferObject *)a->buf)->ptr + (unsigned long long)(a->offset * ITEMSIZE[a->code]));
  d->dtype = a->code;
  d->ndim = o->ndim;
  for (int i = 0; i < o->ndim; i++) {
    d->shape[i] = o->shape[i];
    d->strides[i] = o->strides[i];
  }
  d->is_scalar = 0;
  d->_pad = 0;
  d->scalar_i = 0;
  d->scalar_f = 0.0;
}

/* operand broadcast to (nd, shape): stride 0 on stretched axes. The shapes are known to be compatible. */
static inline void operand_desc(const Operand *o, int nd, const int64_t *shape, mdhip_array *d) {
  if (!o->is_array) {
    memset(d, 0, sizeof(*d));
    d->dtype = o->code;
    d->is_scalar = 1;
    d->scalar_i = o->si;
    d->scalar_f = o->sf;
    return;
  }
  array_desc(o, d);
  if (o->ndim == nd) {
    for (int i = 0; i < nd; i++)
      if (o->shape[i] != shape[i]) d->strides[i] = 0;
  } else {
    int lead = nd - o->ndim;
    for (int i = 0; i < lead; i++) d->strides[i] = 0;
    for (int i = 0; i < o->ndim; i++) d->strides[lead + i] = (o->shape[i] == shape[lead + i]) ? o->strides[i] : 0;
  }
  d->ndim = nd;
  for (int i = 0; i < nd; i++) d->shape[i] = shape[i];
}

/* A fresh C-contiguous array of (nd, shape) and dtype `code`; `like` (may be NULL) is an operand whose shape / strides tuples
 * are reused when they are the result's. Fills `d`. NULL without an exception when the allocation failed. */
static ArrayObject *result_array(int nd, const int64_t *shape, int code, const Operand *like, const Operand *like2, mdhip_array *d) {
  int64_t cst[MDHIP_MAX_NDIM], n = 1;
  for (int i = nd - 1; i >= 0; i--) {
    cst[i] = n;
    n *= shape[i];
  }
  PyObject *shape_t = NULL, *strides_t = NULL;
  const Operand *cands[2] = {like, like2};
  for (int c = 0; c < 2 && !shape_t; c++) {
    const Operand *o = cands[c];
    if (!o || !o->is_array || o->ndim != nd) continue;
    int same = 1, contig = 1;
    for (int i = 0; i < nd; i++) {
      same &= (o->shape[i] == shape[i]);
      contig &= (o->strides[i] == cst[i]);
    }
    if (!same) continue;
    shape_t = o->arr->shape;
    Py_INCREF(shape_t);
    if (contig) {
      strides_t = o->arr->strides;
      Py_INCREF(strides_t);
    }
  }
  if (!shape_t) {
    shape_t = PyTuple_New(nd);
    if (!shape_t) return NULL;
    for (int i = 0; i < nd; i++) PyTuple_SET_ITEM(shape_t, i, PyLong_FromLongLong(shape[i]));
  }
  if (!strides_t) {
    strides_t = PyTuple_New(nd);
    if (!strides_t) {
      Py_DECREF(shape_t);
      return NULL;
    }
    for (int i = 0; i < nd; i++) PyTuple_SET_ITEM(strides_t, i, PyLong_FromLongLong(cst[i]));
  }
  BufferObject *buf = buffer_create((Py_ssize_t)(n * ITEMSIZE[code]), 1);
  if (!buf) {
    Py_DECREF(shape_t);
    Py_DECREF(strides_t);
    return NULL;
  }
  ArrayObject *r = (ArrayObject *)G.array_type->tp_alloc(G.array_type, 0);
  if (!r) {
    Py_DECREF(shape_t);
    Py_DECREF(strides_t);
    Py_DECREF(buf);
    return NULL;
  }
  r->buf = (PyObject *)buf;
  r->offset = 0;
  r->shape = shape_t;
  r->strides = strides_t;
  r->dtype = G.dtypes[code];
  Py_INCREF(r->dtype);
  r->code = code;
  d->data = (void *)(uintptr_t)buf->ptr;
  d->dtype = code;
  d->ndim = nd;
  for (int i = 0; i < nd; i++) {
    d->shape[i] = shape[i];
    d->strides[i] = cst[i];
  }
  d->is_scalar = 0;
  d->_pad = 0;
  d->scalar_i = 0;
  d->scalar_f = 0.0;
  return r;
}

/* result dtype of op(float, float) under NumPy's loops: the float itself, bool for comparisons, -1: not served here */
static inline int binary_out_code(int op, int cdt) {
  if (op >= MDHIP_B_ADD && op <= MDHIP_B_MINIMUM) return cdt;
  if (op >= MDHIP_B_EQ && op <= MDHIP_B_GE) return MDHIP_BOOL;
  return -1;
}

static inline int unary_out_code(int op, int cdt) {
  if (op >= MDHIP_U_ABS && op <= MDHIP_U_SQRT) return cdt;
  if (op == MDHIP_U_ISNAN || op == MDHIP_U_LOGICAL_NOT) return MDHIP_BOOL;
  return -1;
}

/* the transposed-operand rule of ndarray._straighten: such operands take a tiled copy first (Python decides) */
static inline int needs_straighten(const Operand *o) {
  if (!o->is_array || o->ndim < 2) return 0;
  int64_t last = o->strides[o->ndim - 1];
  return !(last == 1 || last == 0);
}

/* NULL without an exception: not served (repeat in Python) */
static PyObject *binary_route(int op, PyObject *pa, PyObject *pb) {
  Operand a, b;
  if (!parse_operand_ex(pa, &a, 1) || !parse_operand_ex(pb, &b, 1)) return NULL;
  if (!a.is_array && !b.is_array) return NULL;
  int cdt;
  if (a.is_array && b.is_array) {
    /* a bool array next to a float array computes in the float type (NumPy's loop for (f, ?): the grad * mask of a relu) */
    if (a.code == MDHIP_BOOL && b.code != MDHIP_BOOL) cdt = b.code;
    else if (b.code == MDHIP_BOOL && a.code != MDHIP_BOOL) cdt = a.code;
    else if (a.code != b.code) return NULL;
    else cdt = a.code;
  } else {
    cdt = a.is_array ? a.code : b.code;
  }
  if (cdt == MDHIP_BOOL) return NULL;   /* bool with bool / with a Python scalar: NumPy's own rules, in Python */
  int odt = binary_out_code(op, cdt);
  if (odt < 0) return NULL;
  if (needs_straighten(&a) || needs_straighten(&b)) return NULL;
  int nd;
  int64_t shape[MDHIP_MAX_NDIM];
  if (a.is_array && b.is_array) {
    nd = a.ndim > b.ndim ? a.ndim : b.ndim;
    for (int i = 1; i <= nd; i++) {
      int64_t x = i <= a.ndim ? a.shape[a.ndim - i] : 1, y = i <= b.ndim ? b.shape[b.ndim - i] : 1;
      if (x == y || y == 1)
        shape[nd - i] = x;
      else if (x == 1)
        shape[nd - i] = y;
      else
        return NULL; /* Python raises NumPy's broadcast error */
    }
  } else {
    const Operand *o = a.is_array ? &a : &b;
    nd = o->ndim;
    for (int i = 0; i < nd; i++) shape[i] = o->shape[i];
  }
  mdhip_array da, db, dr;
  operand_desc(&a, nd, shape, &da);
  operand_desc(&b, nd, shape, &db);
  ArrayObject *r = result_array(nd, shape, odt, &a, &b, &dr);
  if (!r) {
    PyErr_Clear();
    return NULL;
  }
  if (G.binary(op, &da, &db, &dr, cdt)) {
    Py_DECREF(r);
    return NULL;
  }
  return (PyObject *)r;
}

static PyObject *unary_route(int op, PyObject *px) {
  Operand x;
  if (!parse_operand(px, &x) || !x.is_array) return NULL;
  int odt = unary_out_code(op, x.code);
  if (odt < 0) return NULL;
  mdhip_array dx, dr;
  array_desc(&x, &dx);
  ArrayObject *r = result_array(x.ndim, x.shape, odt, &x, NULL, &dr);
  if (!r) {
    PyErr_Clear();
    return NULL;
  }
  if (G.unary(op, &dx, &dr)) {
    Py_DECREF(r);
    return NULL;
  }
  return (PyObject *)r;
}

/* a @ b for two 2-D float arrays of one dtype (the C-ABI takes (batch, rows, cols) descriptors) */
static PyObject *matmul_route(PyObject *pa, PyObject *pb) {
  Operand a, b;
  if (!G.matmul || !parse_operand(pa, &a) || !parse_operand(pb, &b)) return NULL;
  if (!a.is_array || !b.is_array || a.code != b.code || a.ndim != 2 || b.ndim != 2) return NULL;
  const int64_t M = a.shape[0], K = a.shape[1], N = b.shape[1];
  if (b.shape[0] != K || K == 0 || M == 0 || N == 0) return NULL;   /* (mismatch: NumPy's message; empty: a fill, both in Python) */
  mdhip_array da, db, dc;
  array_desc(&a, &da);
  array_desc(&b, &db);
  da.ndim = db.ndim = 3;
  da.shape[0] = 1; da.shape[1] = M; da.shape[2] = K; da.strides[0] = 0; da.strides[1] = a.strides[0]; da.strides[2] = a.strides[1];
  db.shape[0] = 1; db.shape[1] = K; db.shape[2] = N; db.strides[0] = 0; db.strides[1] = b.strides[0]; db.strides[2] = b.strides[1];
  const int64_t shape[2] = {M, N};
  ArrayObject *r = result_array(2, shape, a.code, NULL, NULL, &dc);
  if (!r) {
    PyErr_Clear();
    return NULL;
  }
  dc.ndim = 3;
  dc.shape[0] = 1; dc.shape[1] = M; dc.shape[2] = N; dc.strides[0] = M * N; dc.strides[1] = N; dc.strides[2] = 1;
  if (G.matmul(&da, &db, &dc)) {
    Py_DECREF(r);
    return NULL;
  }
  return (PyObject *)r;
}

/* where(cond, x, y): cond a bool array; x, y float arrays of one dtype and / or weak Python scalars (at least one array) */
static PyObject *where_route(PyObject *pc, PyObject *px, PyObject *py) {
  Operand c, x, y;
  if (!G.where || !parse_operand_ex(pc, &c, 1) || !parse_operand(px, &x) || !parse_operand(py, &y)) return NULL;
  if (!c.is_array || c.code != MDHIP_BOOL) return NULL;
  if (!x.is_array && !y.is_array) return NULL;          /* two Python scalars: NumPy's default dtypes, in Python */
  if (x.is_array && y.is_array && x.code != y.code) return NULL;
  const int odt = x.is_array ? x.code : y.code;
  const Operand *arrs[3] = {&c, x.is_array ? &x : NULL, y.is_array ? &y : NULL};
  int nd = 0;
  for (int i = 0; i < 3; i++)
    if (arrs[i] && arrs[i]->ndim > nd) nd = arrs[i]->ndim;
  int64_t shape[MDHIP_MAX_NDIM];
  for (int d = 1; d <= nd; d++) {
    int64_t e = 1;
    for (int i = 0; i < 3; i++) {
      if (!arrs[i] || d > arrs[i]->ndim) continue;
      const int64_t v = arrs[i]->shape[arrs[i]->ndim - d];
      if (v == e || v == 1) continue;
      if (e == 1) e = v;
      else return NULL;   /* Python raises NumPy's broadcast error */
    }
    shape[nd - d] = e;
  }
  int64_t total = 1;
  for (int i = 0; i < nd; i++) total *= shape[i];
  if (total == 0) return NULL;
  mdhip_array dcd, dx, dy, dr;
  operand_desc(&c, nd, shape, &dcd);
  operand_desc(&x, nd, shape, &dx);
  operand_desc(&y, nd, shape, &dy);
  ArrayObject *r = result_array(nd, shape, odt, x.is_array ? &x : &y, y.is_array ? &y : NULL, &dr);
  if (!r) {
    PyErr_Clear();
    return NULL;
  }
  if (G.where(&dcd, &dx, &dy, &dr)) {
    Py_DECREF(r);
    return NULL;
  }
  return (PyObject *)r;
}

/* sum / prod / max / min of a SMALL float array (below the size at which Python re-expresses some forms through other kernels):
 * args (x[, axis]), keywords axis / keepdims (dtype / out only as None) */
#define MD_FAST_REDUCE_MAX (1 << 18)
static int kw_is(PyObject *name, const char *s) { return PyUnicode_CompareWithASCIIString(name, s) == 0; }

static PyObject *reduce_route(int op, PyObject *const *args, Py_ssize_t nargs, PyObject *kwnames) {
  if (nargs < 1 || nargs > 2) return NULL;
  PyObject *axis = nargs == 2 ? args[1] : Py_None;
  int keepdims = 0;
  const Py_ssize_t nkw = kwnames ? PyTuple_GET_SIZE(kwnames) : 0;
  for (Py_ssize_t i = 0; i < nkw; i++) {
    PyObject *name = PyTuple_GET_ITEM(kwnames, i), *v = args[nargs + i];
    if (kw_is(name, "axis")) {
      if (nargs == 2) return NULL;
      axis = v;
    } else if (kw_is(name, "keepdims")) {
      if (v == Py_True) keepdims = 1;
      else if (v != Py_False) return NULL;
    } else if (kw_is(name, "dtype") || kw_is(name, "out")) {
      if (v != Py_None) return NULL;
    } else {
      return NULL;
    }
  }
  Operand x;
  if (!parse_operand(args[0], &x) || !x.is_array || x.ndim == 0) return NULL;
  int64_t total = 1;
  for (int i = 0; i < x.ndim; i++) total *= x.shape[i];
  if (total == 0 || total >= MD_FAST_REDUCE_MAX) return NULL;
  uint32_t mask = 0;
  if (axis == Py_None) {
    mask = (1u << x.ndim) - 1u;
  } else if (PyLong_CheckExact(axis)) {
    long v = PyLong_AsLong(axis);
    if (v == -1 && PyErr_Occurred()) { PyErr_Clear(); return NULL; }
    if (v < -x.ndim || v >= x.ndim) return NULL;
    mask = 1u << (v < 0 ? v + x.ndim : v);
  } else if (PyTuple_CheckExact(axis)) {
    for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(axis); i++) {
      PyObject *e = PyTuple_GET_ITEM(axis, i);
      if (!PyLong_CheckExact(e)) return NULL;
      long v = PyLong_AsLong(e);
      if (v == -1 && PyErr_Occurred()) { PyErr_Clear(); return NULL; }
      if (v < -x.ndim || v >= x.ndim) return NULL;
      const uint32_t bit = 1u << (v < 0 ? v + x.ndim : v);
      if (mask & bit) return NULL;   /* duplicate: NumPy's ValueError, in Python */
      mask |= bit;
    }
  } else {
    return NULL;
  }
  /* the kernel writes the kept-dims form; the returned array has the reduced axes dropped unless keepdims */
  int64_t kshape[MDHIP_MAX_NDIM], fshape[MDHIP_MAX_NDIM];
  int fnd = 0;
  for (int i = 0; i < x.ndim; i++) {
    const int red = (mask >> i) & 1u;
    kshape[i] = red ? 1 : x.shape[i];
    if (keepdims || !red) fshape[fnd++] = kshape[i];
  }
  mdhip_array dx, dr;
  array_desc(&x, &dx);
  ArrayObject *r = result_array(fnd, fshape, x.code, NULL, NULL, &dr);
  if (!r) {
    PyErr_Clear();
    return NULL;
  }
  dr.ndim = x.ndim;
  int64_t acc = 1;
  for (int i = x.ndim - 1; i >= 0; i--) {
    dr.shape[i] = kshape[i];
    dr.strides[i] = acc;
    acc *= kshape[i];
  }
  if (G.reduce(op, &dx, &dr, mask)) {
    Py_DECREF(r);
    return NULL;
  }
  return (PyObject *)r;
}

/* =========================================================================== FastOp */
enum { KIND_UNARY = 1, KIND_BINARY = 2, KIND_MATMUL = 3, KIND_WHERE = 4, KIND_REDUCE = 5 };

typedef struct {
  PyObject_HEAD
  vectorcallfunc vectorcall;
  int kind; /* KIND_* */
  int code;
  PyObject *slow; /* the Python implementation */
  PyObject *name;
  PyObject *dict;
} FastOpObject;

static PyObject *FastOp_vectorcall(PyObject *self_, PyObject *const *args, size_t nargsf, PyObject *kwnames) {
  FastOpObject *self = (FastOpObject *)self_;
  if (G.ops_enabled && !G.lazy) {
    const Py_ssize_t n = PyVectorcall_NARGS(nargsf);
    const int nokw = !kwnames || PyTuple_GET_SIZE(kwnames) == 0;
    PyObject *r = NULL;
    switch (self->kind) {
      case KIND_UNARY: if (n == 1 && nokw) r = unary_route(self->code, args[0]); break;
      case KIND_BINARY: if (n == 2 && nokw) r = binary_route(self->code, args[0], args[1]); break;
      case KIND_MATMUL: if (n == 2 && nokw) r = matmul_route(args[0], args[1]); break;
      case KIND_WHERE: if (n == 3 && nokw) r = where_route(args[0], args[1], args[2]); break;
      case KIND_REDUCE: r = reduce_route(self->code, args, n, kwnames); break;
      default: break;
    }
    if (r) {
      G.served++;
      return r;
    }
    if (PyErr_Occurred()) PyErr_Clear();
  }
  G.passed++;
  return PyObject_Vectorcall(self->slow, args, nargsf, kwnames);
}

static PyObject *FastOp_new(PyTypeObject *type, PyObject *args, PyObject *kw) {
  int kind, code;
  PyObject *slow, *name;
  if (!PyArg_ParseTuple(args, "iiOU", &kind, &code, &slow, &name)) return NULL;
  if (kind < KIND_UNARY || kind > KIND_REDUCE || !PyCallable_Check(slow)) {
    PyErr_SetString(PyExc_TypeError, "FastOp(kind: 1 unary / 2 binary / 3 matmul / 4 where / 5 reduce, code, callable, name)");
    return NULL;
  }
  FastOpObject *f = (FastOpObject *)type->tp_alloc(type, 0);
  if (!f) return NULL;
  f->vectorcall = FastOp_vectorcall;
  f->kind = kind;
  f->code = code;
  Py_INCREF(slow);
  f->slow = slow;
  Py_INCREF(name);
  f->name = name;
  f->dict = NULL;
  return (PyObject *)f;
}

static int FastOp_traverse(FastOpObject *f, visitproc visit, void *arg) {
  Py_VISIT(f->slow);
  Py_VISIT(f->dict);
  return 0;
}

static int FastOp_clear(FastOpObject *f) {
  Py_CLEAR(f->slow);
  Py_CLEAR(f->dict);
  return 0;
}

static void FastOp_dealloc(FastOpObject *f) {
  PyObject_GC_UnTrack(f);
  FastOp_clear(f);
  Py_CLEAR(f->name);
  Py_TYPE(f)->tp_free((PyObject *)f);
}

static PyObject *FastOp_repr(FastOpObject *f) { return PyUnicode_FromFormat("<fast op %U>", f->name); }

static PyMemberDef FastOp_members[] = {
    {"__wrapped__", T_OBJECT, offsetof(FastOpObject, slow), READONLY, "the Python implementation"},
    {"__name__", T_OBJECT, offsetof(FastOpObject, name), READONLY, NULL},
    {"__qualname__", T_OBJECT, offsetof(FastOpObject, name), READONLY, NULL},
    {"code", T_INT, offsetof(FastOpObject, code), READONLY, NULL},
    {NULL}};

static PyTypeObject FastOp_Type = {
    PyVarObject_HEAD_INIT(NULL, 0).tp_name = "minidiff_amd._fastpath.FastOp",
    .tp_basicsize = sizeof(FastOpObject),
    .tp_dealloc = (destructor)FastOp_dealloc,
    .tp_vectorcall_offset = offsetof(FastOpObject, vectorcall),
    .tp_repr = (reprfunc)FastOp_repr,
    .tp_call = PyVectorcall_Call,
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_GC | Py_TPFLAGS_HAVE_VECTORCALL,
    .tp_doc = "FastOp(kind, code, python_implementation, name): one table entry with the C route in front.",
    .tp_traverse = (traverseproc)FastOp_traverse,
    .tp_clear = (inquiry)FastOp_clear,
    .tp_members = FastOp_members,
    .tp_dictoffset = offsetof(FastOpObject, dict),
    .tp_new = FastOp_new,
};

/* =========================================================================== module functions */
/* new_array(shape: tuple[int], dtype: np.dtype, code: int) -> DeviceArray — the result block of an op (ndarray.DeviceArray._new) */
static PyObject *fp_new_array(PyObject *mod, PyObject *const *args, Py_ssize_t nargs) {
  if (nargs != 3 || !PyTuple_CheckExact(args[0])) {
    PyErr_SetString(PyExc_TypeError, "new_array(shape: tuple, dtype, code)");
    return NULL;
  }
  if (ensure_bound() < 0) return NULL;
  long code = PyLong_AsLong(args[2]);
  if (code == -1 && PyErr_Occurred()) return NULL;
  Py_ssize_t nd = PyTuple_GET_SIZE(args[0]);
  int64_t shape[MDHIP_MAX_NDIM], cst[MDHIP_MAX_NDIM], n = 1;
  if (nd > MDHIP_MAX_NDIM || code < 0 || code >= MDHIP_NUM_ALL_DTYPES || !G.array_type) {
    PyErr_SetString(PyExc_ValueError, "new_array: bad rank or dtype code");
    return NULL;
  }
  if (tuple_to_i64(args[0], shape, (int)nd) < 0) {
    PyErr_SetString(PyExc_TypeError, "new_array: shape must be a tuple of int");
    return NULL;
  }
  for (int i = (int)nd - 1; i >= 0; i--) {
    cst[i] = n;
    n *= shape[i];
  }
  PyObject *itemsize = PyObject_GetAttrString(args[1], "itemsize");
  if (!itemsize) return NULL;
  Py_ssize_t isz = PyLong_AsSsize_t(itemsize);
  Py_DECREF(itemsize);
  if (isz == -1 && PyErr_Occurred()) return NULL;
  PyObject *strides_t = PyTuple_New(nd);
  if (!strides_t) return NULL;
  for (int i = 0; i < nd; i++) PyTuple_SET_ITEM(strides_t, i, PyLong_FromLongLong(cst[i]));
  BufferObject *buf = buffer_create((Py_ssize_t)n * isz, 0);
  if (!buf) {
    Py_DECREF(strides_t);
    return NULL;
  }
  ArrayObject *r = (ArrayObject *)G.array_type->tp_alloc(G.array_type, 0);
  if (!r) {
    Py_DECREF(strides_t);
    Py_DECREF(buf);
    return NULL;
  }
  r->buf = (PyObject *)buf;
  r->offset = 0;
  Py_INCREF(args[0]);
  r->shape = args[0];
  r->strides = strides_t;
  Py_INCREF(args[1]);
  r->dtype = args[1];
  r->code = (int)code;
  return (PyObject *)r;
}

static unsigned long long addr_of(PyObject *d, const char *key) {
  PyObject *v = PyDict_GetItemString(d, key);
  if (!v) {
    PyErr_Format(PyExc_KeyError, "bind: missing %s", key);
    return 0;
  }
  return PyLong_AsUnsignedLongLong(v);
}

/* bind(addresses: dict name -> int) — function pointers of the library the process is bound to (re-bound when tests switch) */
static PyObject *fp_bind(PyObject *mod, PyObject *d) {
  if (!PyDict_Check(d)) {
    PyErr_SetString(PyExc_TypeError, "bind(dict)");
    return NULL;
  }
  unsigned long long a = addr_of(d, "mdhip_alloc"), f = addr_of(d, "mdhip_free"), u = addr_of(d, "mdhip_unary"), b = addr_of(d, "mdhip_binary"),
                     r = addr_of(d, "mdhip_reduce"), mm = addr_of(d, "mdhip_matmul"), wh = addr_of(d, "mdhip_where");
  if (PyErr_Occurred()) return NULL;
  if (!a || !f || !u || !b || !r || !mm || !wh) {
    PyErr_SetString(PyExc_ValueError, "bind: null entry point");
    return NULL;
  }
  G.alloc = (alloc_fn)(uintptr_t)a;
  G.release = (free_fn)(uintptr_t)f;
  G.unary = (unary_fn)(uintptr_t)u;
  G.binary = (binary_fn)(uintptr_t)b;
  G.reduce = (reduce_fn)(uintptr_t)r;
  G.matmul = (matmul_fn)(uintptr_t)mm;
  G.where = (where_fn)(uintptr_t)wh;
  Py_RETURN_NONE;
}

/* configure(array_type, dtypes: tuple of 5 np.dtype, dtype_code, raise_status, loader) */
static PyObject *fp_configure(PyObject *mod, PyObject *args) {
  PyObject *t, *dts, *dc, *rs, *ld;
  if (!PyArg_ParseTuple(args, "OO!OOO", &t, &PyTuple_Type, &dts, &dc, &rs, &ld)) return NULL;
  if (!PyType_Check(t) || !PyType_IsSubtype((PyTypeObject *)t, &ArrayBase_Type) || ((PyTypeObject *)t)->tp_basicsize != sizeof(ArrayObject)) {
    PyErr_SetString(PyExc_TypeError, "configure: array type must subclass ArrayBase without adding fields");
    return NULL;
  }
  if (PyTuple_GET_SIZE(dts) != MDHIP_NUM_DTYPES) {
    PyErr_SetString(PyExc_ValueError, "configure: one np.dtype per compute dtype code");
    return NULL;
  }
  Py_INCREF(t);
  Py_XSETREF(G.array_type, (PyTypeObject *)t);
  for (int i = 0; i < MDHIP_NUM_DTYPES; i++) {
    PyObject *x = PyTuple_GET_ITEM(dts, i);
    Py_INCREF(x);
    Py_XSETREF(G.dtypes[i], x);
  }
  Py_INCREF(dc);
  Py_XSETREF(G.dtype_code, dc);
  Py_INCREF(rs);
  Py_XSETREF(G.raise_status, rs);
  Py_INCREF(ld);
  Py_XSETREF(G.loader, ld);
  Py_RETURN_NONE;
}

static PyObject *fp_set_lazy(PyObject *mod, PyObject *v) {
  int f = PyObject_IsTrue(v);
  if (f < 0) return NULL;
  G.lazy = f;
  Py_RETURN_NONE;
}

static PyObject *fp_enable_ops(PyObject *mod, PyObject *v) {
  int f = PyObject_IsTrue(v);
  if (f < 0) return NULL;
  int prev = G.ops_enabled;
  G.ops_enabled = f && G.array_type && G.binary;
  return PyBool_FromLong(prev);
}

static PyObject *fp_stats(PyObject *mod, PyObject *unused) {
  return Py_BuildValue("{s:K,s:K,s:i,s:i}", "served", G.served, "passed", G.passed, "enabled", G.ops_enabled, "lazy", G.lazy);
}

/* time_binary(op, a, b, n) -> seconds for n calls of mdhip_binary on prepared descriptors (one result block, reused): the
 * C-ABI call + launch alone, i.e. the floor under any host route (scripts/host_overhead.py calls). */
static PyObject *fp_time_binary(PyObject *mod, PyObject *args) {
  int op;
  PyObject *pa, *pb;
  Py_ssize_t n;
  if (!PyArg_ParseTuple(args, "iOOn", &op, &pa, &pb, &n)) return NULL;
  if (ensure_bound() < 0) return NULL;
  Operand a, b;
  if (!G.array_type || !parse_operand(pa, &a) || !parse_operand(pb, &b) || !a.is_array || !b.is_array || a.code != b.code || a.ndim != b.ndim ||
      memcmp(a.shape, b.shape, sizeof(int64_t) * a.ndim) || binary_out_code(op, a.code) < 0) {
    PyErr_SetString(PyExc_ValueError, "time_binary: two float arrays of one dtype and shape");
    return NULL;
  }
  mdhip_array da, db, dr;
  operand_desc(&a, a.ndim, a.shape, &da);
  operand_desc(&b, a.ndim, a.shape, &db);
  ArrayObject *r = result_array(a.ndim, a.shape, binary_out_code(op, a.code), &a, &b, &dr);
  if (!r) return PyErr_Occurred() ? NULL : PyErr_NoMemory();
  struct timespec t0, t1;
  int st = 0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (Py_ssize_t i = 0; i < n && !st; i++) st = G.binary(op, &da, &db, &dr, a.code);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  Py_DECREF(r);
  if (st) return raise_status(st);
  return PyFloat_FromDouble((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
}

static PyMethodDef module_methods[] = {
    {"time_binary", fp_time_binary, METH_VARARGS, "time_binary(op, a, b, n) -> seconds for n C-ABI calls on prepared descriptors"},
    {"new_array", (PyCFunction)(void (*)(void))fp_new_array, METH_FASTCALL, "new_array(shape, dtype, code) -> DeviceArray over a fresh block"},
    {"bind", fp_bind, METH_O, "bind({symbol: address}) — C-ABI entry points of the bound library"},
    {"configure", fp_configure, METH_VARARGS, "configure(array_type, dtypes, dtype_code, raise_status, loader)"},
    {"set_lazy", fp_set_lazy, METH_O, "mirror of ndarray's lazy switch (the C route serves eager mode only)"},
    {"enable_ops", fp_enable_ops, METH_O, "switch the C route of FastOp on / off; returns the previous setting"},
    {"stats", fp_stats, METH_NOARGS, "{'served', 'passed', 'enabled', 'lazy'}"},
    {NULL}};

static struct PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "minidiff_amd._fastpath", "C host route of eager backend calls (see csrc/fastpath.c)", -1,
                                        module_methods};

PyMODINIT_FUNC PyInit__fastpath(void) {
  if (PyType_Ready(&Buffer_Type) < 0 || PyType_Ready(&ArrayBase_Type) < 0 || PyType_Ready(&FastOp_Type) < 0) return NULL;
  PyObject *m = PyModule_Create(&module_def);
  if (!m) return NULL;
  Py_INCREF(&Buffer_Type);
  Py_INCREF(&ArrayBase_Type);
  Py_INCREF(&FastOp_Type);
  if (PyModule_AddObject(m, "Buffer", (PyObject *)&Buffer_Type) < 0 || PyModule_AddObject(m, "ArrayBase", (PyObject *)&ArrayBase_Type) < 0 ||
      PyModule_AddObject(m, "FastOp", (PyObject *)&FastOp_Type) < 0)
    return NULL;
  PyModule_AddIntConstant(m, "ABI_DESC_BYTES", (long)sizeof(mdhip_array));
  return m;
}
