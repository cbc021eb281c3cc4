/* _tgxfast: the two ends of the PyO3-shaped list surface in native code (CPython C API, no numpy dependency).
 *
 * The reference's binding takes `list[str]` and returns `list[list[int]]` built in Rust (bindings/python/src/lib.rs:
 * 51-69).  Between those two shapes and the C ABI's packed buffers (include/tgx.h) the Python mirror spent 0.5 s per
 * 64 MiB: one bytes object and one UTF-8 encode per sample on the way in, one int object per token on the way out
 * (profiles/r03/final/python_api_rate.json: 0.12 GB/s).  Here:
 *
 *   pack_strs(list[str]) -> (bytes text, bytes offsets)      UTF-8 of every sample back to back + u64[n + 1] offsets,
 *       encoded straight from the strings' internal 1 / 2 / 4-byte representations by a few host threads with the GIL
 *       released (the strings are immutable and held by references taken first);
 *   rows_from_flat(ids u32 buffer, offs u64 buffer, cache list[int]) -> list[list[int]]
 *       one list per sample whose items are the SHARED int objects of `cache` (cache[id] == id, made once per
 *       tokenizer): a token costs a reference count and a pointer store instead of an allocation — ints are immutable, so
 *       sharing them is invisible (CPython itself shares -5 .. 256).
 *
 * Errors follow str.encode("utf-8"): a lone surrogate raises UnicodeEncodeError.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    int kind;          /* 1, 2, 4 */
    int ascii;
    const void* data;
    Py_ssize_t len;    /* code points */
    uint64_t bytes;    /* UTF-8 bytes */
    int bad;           /* lone surrogate */
} StrView;

static uint64_t utf8_len(const StrView* s, int* bad) {
    const Py_ssize_t n = s->len;
    uint64_t b = 0;
    if (s->kind == 1) {
        if (s->ascii) return (uint64_t)n;
        const uint8_t* p = (const uint8_t*)s->data;
        for (Py_ssize_t i = 0; i < n; i++) b += 1u + (p[i] >> 7);
    } else if (s->kind == 2) {
        const uint16_t* p = (const uint16_t*)s->data;
        for (Py_ssize_t i = 0; i < n; i++) {
            const uint32_t c = p[i];
            if (c >= 0xD800u && c <= 0xDFFFu) *bad = 1;
            b += c < 0x80u ? 1u : (c < 0x800u ? 2u : 3u);
        }
    } else {
        const uint32_t* p = (const uint32_t*)s->data;
        for (Py_ssize_t i = 0; i < n; i++) {
            const uint32_t c = p[i];
            if (c >= 0xD800u && c <= 0xDFFFu) *bad = 1;
            b += c < 0x80u ? 1u : (c < 0x800u ? 2u : (c < 0x10000u ? 3u : 4u));
        }
    }
    return b;
}

static void utf8_put(const StrView* s, uint8_t* out) {
    const Py_ssize_t n = s->len;
    if (s->kind == 1 && s->ascii) {
        memcpy(out, s->data, (size_t)n);
        return;
    }
    for (Py_ssize_t i = 0; i < n; i++) {
        const uint32_t c = s->kind == 1 ? ((const uint8_t*)s->data)[i] : (s->kind == 2 ? ((const uint16_t*)s->data)[i] : ((const uint32_t*)s->data)[i]);
        if (c < 0x80u) {
            *out++ = (uint8_t)c;
        } else if (c < 0x800u) {
            *out++ = (uint8_t)(0xC0u | (c >> 6));
            *out++ = (uint8_t)(0x80u | (c & 0x3Fu));
        } else if (c < 0x10000u) {
            *out++ = (uint8_t)(0xE0u | (c >> 12));
            *out++ = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu));
            *out++ = (uint8_t)(0x80u | (c & 0x3Fu));
        } else {
            *out++ = (uint8_t)(0xF0u | (c >> 18));
            *out++ = (uint8_t)(0x80u | ((c >> 12) & 0x3Fu));
            *out++ = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu));
            *out++ = (uint8_t)(0x80u | (c & 0x3Fu));
        }
    }
}

typedef struct {
    StrView* v;
    Py_ssize_t lo, hi;
    const uint64_t* offs;
    uint8_t* out;
    int phase;  /* 0: lengths, 1: encode */
} Job;

static void* job_run(void* arg) {
    Job* j = (Job*)arg;
    if (j->phase == 0) {
        for (Py_ssize_t i = j->lo; i < j->hi; i++) {
            int bad = 0;
            j->v[i].bytes = utf8_len(&j->v[i], &bad);
            j->v[i].bad = bad;
        }
    } else {
        for (Py_ssize_t i = j->lo; i < j->hi; i++) utf8_put(&j->v[i], j->out + j->offs[i]);
    }
    return NULL;
}

static void run_parallel(StrView* v, Py_ssize_t n, const uint64_t* offs, uint8_t* out, int phase, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if (n < 256) threads = 1;
    Job jobs[16];
    pthread_t th[16];
    /* slices of about equal numbers of code points */
    uint64_t total = 0;
    for (Py_ssize_t i = 0; i < n; i++) total += (uint64_t)v[i].len + 8u;
    Py_ssize_t at = 0;
    uint64_t acc = 0;
    for (int t = 0; t < threads; t++) {
        jobs[t].v = v;
        jobs[t].offs = offs;
        jobs[t].out = out;
        jobs[t].phase = phase;
        jobs[t].lo = at;
        const uint64_t want = total / (uint64_t)threads * (uint64_t)(t + 1);
        while (at < n && (t + 1 == threads || acc < want)) acc += (uint64_t)v[at++].len + 8u;
        jobs[t].hi = at;
    }
    int started = 0;
    for (int t = 1; t < threads; t++) {
        if (pthread_create(&th[t], NULL, job_run, &jobs[t]) != 0) break;
        started = t;
    }
    job_run(&jobs[0]);
    for (int t = started + 1; t < threads; t++) job_run(&jobs[t]);  /* threads that could not start: inline */
    for (int t = 1; t <= started; t++) pthread_join(th[t], NULL);
}

static PyObject* pack_strs(PyObject* self, PyObject* args) {
    PyObject* seq_in;
    int threads = 8;
    if (!PyArg_ParseTuple(args, "O|i", &seq_in, &threads)) return NULL;
    PyObject* seq = PySequence_Fast(seq_in, "pack_strs: expected a sequence of str");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject** items = PySequence_Fast_ITEMS(seq);
    StrView* v = (StrView*)PyMem_Malloc(sizeof(StrView) * (size_t)(n ? n : 1));
    PyObject** held = (PyObject**)PyMem_Malloc(sizeof(PyObject*) * (size_t)(n ? n : 1));
    if (!v || !held) {
        PyMem_Free(v);
        PyMem_Free(held);
        Py_DECREF(seq);
        return PyErr_NoMemory();
    }
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* o = items[i];
        if (!PyUnicode_Check(o) || PyUnicode_READY(o) != 0) {
            for (Py_ssize_t k = 0; k < i; k++) Py_DECREF(held[k]);
            PyMem_Free(v);
            PyMem_Free(held);
            Py_DECREF(seq);
            if (!PyErr_Occurred()) PyErr_Format(PyExc_TypeError, "pack_strs: item %zd is not a str", i);
            return NULL;
        }
        Py_INCREF(o);
        held[i] = o;
        v[i].kind = (int)PyUnicode_KIND(o);
        v[i].ascii = PyUnicode_IS_ASCII(o) ? 1 : 0;
        v[i].data = PyUnicode_DATA(o);
        v[i].len = PyUnicode_GET_LENGTH(o);
        v[i].bytes = 0;
        v[i].bad = 0;
    }
    PyObject* offs_obj = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)(sizeof(uint64_t) * (size_t)(n + 1)));
    if (!offs_obj) goto fail;
    uint64_t* offs = (uint64_t*)PyBytes_AS_STRING(offs_obj);
    Py_BEGIN_ALLOW_THREADS
    run_parallel(v, n, NULL, NULL, 0, threads);
    Py_END_ALLOW_THREADS
    offs[0] = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
        if (v[i].bad) {
            Py_DECREF(offs_obj);
            /* the error str.encode raises, with its message */
            PyObject* r = PyUnicode_AsUTF8String(held[i]);
            Py_XDECREF(r);
            if (!PyErr_Occurred()) PyErr_SetString(PyExc_UnicodeEncodeError, "surrogates not allowed");
            goto fail;
        }
        offs[i + 1] = offs[i] + v[i].bytes;
    }
    {
        PyObject* text_obj = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)offs[n]);
        if (!text_obj) {
            Py_DECREF(offs_obj);
            goto fail;
        }
        uint8_t* out = (uint8_t*)PyBytes_AS_STRING(text_obj);
        Py_BEGIN_ALLOW_THREADS
        run_parallel(v, n, offs, out, 1, threads);
        Py_END_ALLOW_THREADS
        for (Py_ssize_t i = 0; i < n; i++) Py_DECREF(held[i]);
        PyMem_Free(v);
        PyMem_Free(held);
        Py_DECREF(seq);
        return Py_BuildValue("(NN)", text_obj, offs_obj);
    }
fail:
    for (Py_ssize_t i = 0; i < n; i++) Py_DECREF(held[i]);
    PyMem_Free(v);
    PyMem_Free(held);
    Py_DECREF(seq);
    return NULL;
}

static PyObject* rows_from_flat(PyObject* self, PyObject* args) {
    Py_buffer ids, offs;
    PyObject* cache;
    if (!PyArg_ParseTuple(args, "y*y*O", &ids, &offs, &cache)) return NULL;
    PyObject* out = NULL;
    PyObject* cache_fast = NULL;
    if (offs.len < 8 || offs.len % 8 != 0 || ids.len % 4 != 0) {
        PyErr_SetString(PyExc_ValueError, "rows_from_flat: ids must be uint32, offsets uint64 with at least one entry");
        goto done;
    }
    {
        const uint32_t* id = (const uint32_t*)ids.buf;
        const uint64_t* of = (const uint64_t*)offs.buf;
        const Py_ssize_t n = offs.len / 8 - 1;
        const uint64_t T = (uint64_t)(ids.len / 4);
        PyObject** citems = NULL;
        Py_ssize_t csize = 0;
        if (cache != Py_None) {
            cache_fast = PySequence_Fast(cache, "rows_from_flat: cache must be a sequence of int");
            if (!cache_fast) goto done;
            citems = PySequence_Fast_ITEMS(cache_fast);
            csize = PySequence_Fast_GET_SIZE(cache_fast);
        }
        out = PyList_New(n);
        if (!out) goto done;
        for (Py_ssize_t i = 0; i < n; i++) {
            const uint64_t a = of[i], b = of[i + 1];
            if (b < a || b > T) {
                PyErr_SetString(PyExc_ValueError, "rows_from_flat: offsets out of range");
                Py_CLEAR(out);
                goto done;
            }
            PyObject* row = PyList_New((Py_ssize_t)(b - a));
            if (!row) {
                Py_CLEAR(out);
                goto done;
            }
            PyList_SET_ITEM(out, i, row);
            for (uint64_t k = a; k < b; k++) {
                const uint32_t t = id[k];
                PyObject* o;
                if (k + 16 < T) {  /* the shared object's header (its reference count) 16 tokens ahead */
                    const uint32_t tn = id[k + 16];
                    if ((Py_ssize_t)tn < csize) __builtin_prefetch(citems[tn], 1, 1);
                }
                if ((Py_ssize_t)t < csize) {
                    o = citems[t];
                    Py_INCREF(o);
                } else {
                    o = PyLong_FromUnsignedLong((unsigned long)t);
                    if (!o) {
                        Py_CLEAR(out);
                        goto done;
                    }
                }
                PyList_SET_ITEM(row, (Py_ssize_t)(k - a), o);
            }
        }
    }
done:
    Py_XDECREF(cache_fast);
    PyBuffer_Release(&ids);
    PyBuffer_Release(&offs);
    return out;
}

static PyMethodDef methods[] = {
    {"pack_strs", pack_strs, METH_VARARGS, "pack_strs(list[str], threads=8) -> (utf-8 bytes, uint64 offsets as bytes)"},
    {"rows_from_flat", rows_from_flat, METH_VARARGS, "rows_from_flat(ids u32 buffer, offs u64 buffer, cache list[int] | None) -> list[list[int]]"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_tgxfast", "native ends of the list[str] -> list[list[int]] surface", -1, methods};
PyMODINIT_FUNC PyInit__tgxfast(void) { return PyModule_Create(&moddef); }
