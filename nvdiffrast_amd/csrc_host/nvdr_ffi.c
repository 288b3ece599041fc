/* nvdr_ffi.c -- compiled call layer between Python and the C ABI of include/nvdr_hip.h.
 *
 * The reference binds its operators with pybind11 (csrc/torch/torch_bindings.cpp:43-71): a call costs a fraction of a
 * microsecond on the host.  This package's boundary is a plain C ABI reached from Python; through ctypes a call with
 * twenty arguments costs 3-5 us of argument conversion, four to six times per step -- a tenth of the host time of a step
 * that is host-bound at small batches (BASELINE config 2).  This module is the same binding, compiled: `bind(address,
 * signature)` returns a callable that converts its arguments by the signature string and calls the entry point directly.
 *
 *   signature characters: p = pointer (int or None), i = int, n = size_t, L = long long
 *   result character (first): i = int, n = size_t, v = void
 *
 * Every parameter of the C ABI is an integer or a pointer, so on x86-64 (System V) each travels in one 64-bit slot; the
 * entry point is called through a prototype of N 64-bit integers.  No torch types, no device code: plain C, built with gcc.
 * nvdiffrast_amd/_capi.py falls back to ctypes when the module has not been built. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define NVDR_FFI_MAX_ARGS 28

typedef struct {
    PyObject_HEAD
    void* fn;
    int nargs;
    char ret;
    char sig[NVDR_FFI_MAX_ARGS + 1];
} BoundFn;

typedef uint64_t u64;
#define A(i) a[i]

static u64 call_n(void* fn, int n, const u64* a)
{
    switch (n) {
    case 0: return ((u64 (*)(void))fn)();
    case 1: return ((u64 (*)(u64))fn)(A(0));
    case 2: return ((u64 (*)(u64, u64))fn)(A(0), A(1));
    case 3: return ((u64 (*)(u64, u64, u64))fn)(A(0), A(1), A(2));
    case 4: return ((u64 (*)(u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3));
    case 5: return ((u64 (*)(u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4));
    case 6: return ((u64 (*)(u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5));
    case 7: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6));
    case 8: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7));
    case 9: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8));
    case 10: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9));
    case 11: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10));
    case 12: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11));
    case 13: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12));
    case 14: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13));
    case 15: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14));
    case 16: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15));
    case 17: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16));
    case 18: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17));
    case 19: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18));
    case 20: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19));
    case 21: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20));
    case 22: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20), A(21));
    case 23: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20), A(21), A(22));
    case 24: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20), A(21), A(22), A(23));
    case 25: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20), A(21), A(22), A(23), A(24));
    case 26: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20), A(21), A(22), A(23), A(24), A(25));
    case 27: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20), A(21), A(22), A(23), A(24), A(25), A(26));
    default: return ((u64 (*)(u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64, u64))fn)(A(0), A(1), A(2), A(3), A(4), A(5), A(6), A(7), A(8), A(9), A(10), A(11), A(12), A(13), A(14), A(15), A(16), A(17), A(18), A(19), A(20), A(21), A(22), A(23), A(24), A(25), A(26), A(27));
    }
}

static PyObject* bound_call(PyObject* self_, PyObject* const* args, size_t nargsf, PyObject* kwnames)
{
    BoundFn* self = (BoundFn*)self_;
    const Py_ssize_t n = PyVectorcall_NARGS(nargsf);
    if (kwnames && PyTuple_GET_SIZE(kwnames) > 0) { PyErr_SetString(PyExc_TypeError, "nvdr_ffi: no keyword arguments"); return NULL; }
    if (n != self->nargs) { PyErr_Format(PyExc_TypeError, "nvdr_ffi: %d arguments expected, %zd given", self->nargs, n); return NULL; }
    u64 a[NVDR_FFI_MAX_ARGS];
    Py_buffer views[4]; int nviews = 0;                       /* host arrays passed by object (ctypes arrays): held until the call returns */
    PyObject* result = NULL;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* o = args[i];
        const char c = self->sig[i];
        if (o == Py_None) {
            if (c != 'p') { PyErr_Format(PyExc_TypeError, "nvdr_ffi: argument %zd: None for a non-pointer", i); goto done; }
            a[i] = 0;
        } else if (c == 'p' || c == 'n') {
            if (PyLong_Check(o)) {
                const unsigned long long v = PyLong_AsUnsignedLongLong(o);
                if (v == (unsigned long long)-1 && PyErr_Occurred()) goto done;
                a[i] = v;
            } else if (PyIndex_Check(o)) {
                /* an integer-like object (a numpy integer scalar): its VALUE -- never, through the branch below, the address of
                   the host memory that holds it (ctypes raised TypeError here; a device pointer must not silently
                   become a host address) */
                PyObject* idx = PyNumber_Index(o);
                if (!idx) goto done;
                const unsigned long long v = PyLong_AsUnsignedLongLong(idx);
                Py_DECREF(idx);
                if (v == (unsigned long long)-1 && PyErr_Occurred()) goto done;
                a[i] = v;
            } else {
                /* a host array (ctypes array, bytes-like): its buffer's address */
                if (c != 'p' || nviews == 4 || PyObject_GetBuffer(o, &views[nviews], PyBUF_SIMPLE) != 0) {
                    if (!PyErr_Occurred()) PyErr_Format(PyExc_TypeError, "nvdr_ffi: argument %zd: an int, None or a buffer expected", i);
                    goto done;
                }
                a[i] = (u64)(uintptr_t)views[nviews++].buf;
            }
        } else {
            const long long v = PyLong_AsLongLong(o);
            if (v == -1 && PyErr_Occurred()) goto done;
            a[i] = (c == 'i') ? (u64)(int64_t)(int32_t)v : (u64)v;
            if (c == 'i' && v != (long long)(int32_t)v) { PyErr_Format(PyExc_OverflowError, "nvdr_ffi: argument %zd does not fit an int", i); goto done; }
        }
    }
    {
        /* the arguments are plain integers now and the buffer views are held until `done`: the native call runs without the
           GIL, as it did under ctypes.CDLL (one thread per GPU, autograd's worker threads) */
        u64 r;
        Py_BEGIN_ALLOW_THREADS
        r = call_n(self->fn, (int)n, a);
        Py_END_ALLOW_THREADS
        if (self->ret == 'v') { result = Py_None; Py_INCREF(result); }
        else if (self->ret == 'n') result = PyLong_FromUnsignedLongLong(r);
        else result = PyLong_FromLong((long)(int32_t)r);
    }
done:
    for (int v = 0; v < nviews; v++) PyBuffer_Release(&views[v]);
    return result;
}

static PyTypeObject BoundFnType = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "_nvdr_ffi.BoundFn",
    .tp_basicsize = sizeof(BoundFn),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_HAVE_VECTORCALL,
    .tp_call = PyVectorcall_Call,
    .tp_new = NULL,
};

typedef struct { BoundFn base; vectorcallfunc vc; } BoundFnVC;

static PyObject* ffi_bind(PyObject* mod, PyObject* args)
{
    unsigned long long addr; const char* sig;
    if (!PyArg_ParseTuple(args, "Ks", &addr, &sig)) return NULL;
    const size_t len = strlen(sig);
    if (len < 1 || len - 1 > NVDR_FFI_MAX_ARGS || !strchr("inv", sig[0])) { PyErr_SetString(PyExc_ValueError, "nvdr_ffi: bad signature"); return NULL; }
    for (size_t i = 1; i < len; i++) if (!strchr("pinL", sig[i])) { PyErr_SetString(PyExc_ValueError, "nvdr_ffi: bad signature character"); return NULL; }
    BoundFnVC* f = PyObject_New(BoundFnVC, &BoundFnType);
    if (!f) return NULL;
    f->base.fn = (void*)(uintptr_t)addr; f->base.nargs = (int)len - 1; f->base.ret = sig[0];
    memcpy(f->base.sig, sig + 1, len); f->vc = bound_call;
    return (PyObject*)f;
}

static PyMethodDef methods[] = {
    {"bind", ffi_bind, METH_VARARGS, "bind(address, signature) -> callable: signature = result char + one char per argument (p i n L)"},
    {NULL, NULL, 0, NULL}
};

static struct PyModuleDef moddef = { PyModuleDef_HEAD_INIT, "_nvdr_ffi", "compiled call layer for the C ABI of include/nvdr_hip.h", -1, methods };

PyMODINIT_FUNC PyInit__nvdr_ffi(void)
{
    BoundFnType.tp_basicsize = sizeof(BoundFnVC);
    BoundFnType.tp_vectorcall_offset = offsetof(BoundFnVC, vc);
    if (PyType_Ready(&BoundFnType) < 0) return NULL;
    return PyModule_Create(&moddef);
}
