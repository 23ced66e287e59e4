// list[str] -> packed UTF-8 + offsets, packed token ids + offsets -> list[list[int]]: the two ends of Encoding.encode_ordinary_batch /
// encode_batch (reference: PyO3 turns Vec<Vec<Rank>> into lists and &str into UTF-8 at src/py.rs:29-49; tiktoken/core.py:164-206 maps
// the single-text methods over a thread pool).  In Python the two ends cost more than the GPU's work in between: str.encode per text,
// b"".join, a numpy slice and tolist() per document.  Here: one pass for the sizes, one for the bytes (lone surrogates become U+FFFD and
// surrogate pairs one char, as the reference's repair path does, core.py:79,135), and one loop that builds the lists.
// A CPython extension (gcc, no GPU code): tiktoken_amd/_tk_marshal*.so.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

// next scalar value of a str's code units from index *i (kinds 2 and 4 may hold surrogates): pairs are joined, lone ones replaced
static inline uint32_t next_scalar(int kind, const void* data, Py_ssize_t n, Py_ssize_t* i) {
    uint32_t c = (uint32_t)PyUnicode_READ(kind, data, *i);
    ++*i;
    if (c >= 0xD800u && c <= 0xDFFFu) {
        if (c <= 0xDBFFu && *i < n) {
            const uint32_t d = (uint32_t)PyUnicode_READ(kind, data, *i);
            if (d >= 0xDC00u && d <= 0xDFFFu) {
                ++*i;
                return 0x10000u + ((c - 0xD800u) << 10) + (d - 0xDC00u);
            }
        }
        return 0xFFFDu;
    }
    return c;
}
static inline size_t utf8_len(uint32_t c) { return c < 0x80u ? 1 : (c < 0x800u ? 2 : (c < 0x10000u ? 3 : 4)); }
static inline uint8_t* utf8_put(uint8_t* p, uint32_t c) {
    if (c < 0x80u) {
        *p++ = (uint8_t)c;
    } else if (c < 0x800u) {
        *p++ = (uint8_t)(0xC0u | (c >> 6));
        *p++ = (uint8_t)(0x80u | (c & 0x3Fu));
    } else if (c < 0x10000u) {
        *p++ = (uint8_t)(0xE0u | (c >> 12));
        *p++ = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu));
        *p++ = (uint8_t)(0x80u | (c & 0x3Fu));
    } else {
        *p++ = (uint8_t)(0xF0u | (c >> 18));
        *p++ = (uint8_t)(0x80u | ((c >> 12) & 0x3Fu));
        *p++ = (uint8_t)(0x80u | ((c >> 6) & 0x3Fu));
        *p++ = (uint8_t)(0x80u | (c & 0x3Fu));
    }
    return p;
}
static size_t str_utf8_size(PyObject* s) {
    const Py_ssize_t n = PyUnicode_GET_LENGTH(s);
    if (PyUnicode_IS_ASCII(s)) return (size_t)n;
    const int kind = PyUnicode_KIND(s);
    const void* data = PyUnicode_DATA(s);
    size_t total = 0;
    if (kind == PyUnicode_1BYTE_KIND) {
        const uint8_t* b = (const uint8_t*)data;
        Py_ssize_t i = 0;
        for (; i + 8 <= n; i += 8) {  // (Latin-1: a byte per char, two for the upper half)
            uint64_t w;
            memcpy(&w, b + i, 8);
            total += 8u + (size_t)__builtin_popcountll(w & 0x8080808080808080ull);
        }
        for (; i < n; ++i) total += 1u + (b[i] >> 7);
        return total;
    }
    // (wider kinds: mostly ASCII with a few chars beyond Latin-1 is the common case -- runs of ASCII code units are counted four / two at a time)
    if (kind == PyUnicode_2BYTE_KIND) {
        const uint16_t* u = (const uint16_t*)data;
        Py_ssize_t i = 0;
        while (i < n) {
            if (i + 4 <= n) {
                uint64_t w;
                memcpy(&w, u + i, 8);
                if (!(w & 0xFF80FF80FF80FF80ull)) {
                    total += 4;
                    i += 4;
                    continue;
                }
            }
            total += utf8_len(next_scalar(kind, data, n, &i));
        }
        return total;
    }
    {
        const uint32_t* u = (const uint32_t*)data;
        Py_ssize_t i = 0;
        while (i < n) {
            if (i + 2 <= n) {
                uint64_t w;
                memcpy(&w, u + i, 8);
                if (!(w & 0xFFFFFF80FFFFFF80ull)) {
                    total += 2;
                    i += 2;
                    continue;
                }
            }
            total += utf8_len(next_scalar(kind, data, n, &i));
        }
    }
    return total;
}
static uint8_t* str_utf8_write(PyObject* s, uint8_t* p) {
    const Py_ssize_t n = PyUnicode_GET_LENGTH(s);
    const void* data = PyUnicode_DATA(s);
    if (PyUnicode_IS_ASCII(s)) {
        memcpy(p, data, (size_t)n);
        return p + n;
    }
    const int kind = PyUnicode_KIND(s);
    if (kind == PyUnicode_1BYTE_KIND) {
        const uint8_t* b = (const uint8_t*)data;
        Py_ssize_t i = 0;
        while (i < n) {
            if (i + 8 <= n) {  // eight ASCII bytes at once
                uint64_t w;
                memcpy(&w, b + i, 8);
                if (!(w & 0x8080808080808080ull)) {
                    memcpy(p, &w, 8);
                    p += 8;
                    i += 8;
                    continue;
                }
            }
            p = utf8_put(p, b[i++]);
        }
        return p;
    }
    if (kind == PyUnicode_2BYTE_KIND) {
        const uint16_t* u = (const uint16_t*)data;
        Py_ssize_t i = 0;
        while (i < n) {
            if (i + 4 <= n) {
                uint64_t w;
                memcpy(&w, u + i, 8);
                if (!(w & 0xFF80FF80FF80FF80ull)) {
                    p[0] = (uint8_t)w;
                    p[1] = (uint8_t)(w >> 16);
                    p[2] = (uint8_t)(w >> 32);
                    p[3] = (uint8_t)(w >> 48);
                    p += 4;
                    i += 4;
                    continue;
                }
            }
            p = utf8_put(p, next_scalar(kind, data, n, &i));
        }
        return p;
    }
    {
        const uint32_t* u = (const uint32_t*)data;
        Py_ssize_t i = 0;
        while (i < n) {
            if (i + 2 <= n) {
                uint64_t w;
                memcpy(&w, u + i, 8);
                if (!(w & 0xFFFFFF80FFFFFF80ull)) {
                    p[0] = (uint8_t)w;
                    p[1] = (uint8_t)(w >> 32);
                    p += 2;
                    i += 2;
                    continue;
                }
            }
            p = utf8_put(p, next_scalar(kind, data, n, &i));
        }
    }
    return p;
}

// pack(texts) -> (blob: bytes, offsets: bytes holding len(texts) + 1 uint64)
static PyObject* tkm_pack(PyObject* self, PyObject* arg) {
    (void)self;
    PyObject* seq = PySequence_Fast(arg, "expected a sequence of str");
    if (!seq) return NULL;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    PyObject** items = PySequence_Fast_ITEMS(seq);
    PyObject* offs = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)((size_t)(n + 1) * 8u));
    if (!offs) {
        Py_DECREF(seq);
        return NULL;
    }
    uint64_t* off = (uint64_t*)PyBytes_AS_STRING(offs);
    uint64_t total = 0;
    off[0] = 0;
    for (Py_ssize_t k = 0; k < n; ++k) {
        if (!PyUnicode_Check(items[k])) {
            PyErr_Format(PyExc_TypeError, "expected str, got %.80s (item %zd)", Py_TYPE(items[k])->tp_name, k);
            Py_DECREF(offs);
            Py_DECREF(seq);
            return NULL;
        }
        if (PyUnicode_READY(items[k]) < 0) {
            Py_DECREF(offs);
            Py_DECREF(seq);
            return NULL;
        }
        total += str_utf8_size(items[k]);
        off[k + 1] = total;
    }
    PyObject* blob = PyBytes_FromStringAndSize(NULL, (Py_ssize_t)total);
    if (!blob) {
        Py_DECREF(offs);
        Py_DECREF(seq);
        return NULL;
    }
    uint8_t* p = (uint8_t*)PyBytes_AS_STRING(blob);
    for (Py_ssize_t k = 0; k < n; ++k) p = str_utf8_write(items[k], p);
    Py_DECREF(seq);
    PyObject* r = PyTuple_Pack(2, blob, offs);
    Py_DECREF(blob);
    Py_DECREF(offs);
    return r;
}

// unpack(tokens, tok_off) -> list[list[int]]   (tokens: buffer of uint32, tok_off: buffer of n + 1 uint64, non-decreasing)
static PyObject* tkm_unpack(PyObject* self, PyObject* args) {
    (void)self;
    Py_buffer tb, ob;
    if (!PyArg_ParseTuple(args, "y*y*", &tb, &ob)) return NULL;
    PyObject* out = NULL;
    if (ob.len < 8 || ob.len % 8 || tb.len % 4) {
        PyErr_SetString(PyExc_ValueError, "tokens must be uint32, tok_off must hold n + 1 uint64");
        goto done;
    }
    {
        const uint32_t* tok = (const uint32_t*)tb.buf;
        const uint64_t* off = (const uint64_t*)ob.buf;
        const Py_ssize_t n = ob.len / 8 - 1;
        const uint64_t nt = (uint64_t)tb.len / 4;
        for (Py_ssize_t d = 0; d < n; ++d)
            if (off[d + 1] < off[d] || off[d + 1] > nt) {
                PyErr_SetString(PyExc_ValueError, "tok_off must be non-decreasing and end within tokens");
                goto done;
            }
        out = PyList_New(n);
        if (!out) goto done;
        for (Py_ssize_t d = 0; d < n; ++d) {
            const uint64_t a = off[d], b = off[d + 1];
            PyObject* l = PyList_New((Py_ssize_t)(b - a));
            if (!l) {
                Py_CLEAR(out);
                goto done;
            }
            PyList_SET_ITEM(out, d, l);
            for (uint64_t i = a; i < b; ++i) {
                PyObject* v = PyLong_FromUnsignedLong(tok[i]);
                if (!v) {
                    Py_CLEAR(out);
                    goto done;
                }
                PyList_SET_ITEM(l, (Py_ssize_t)(i - a), v);
            }
        }
    }
done:
    PyBuffer_Release(&tb);
    PyBuffer_Release(&ob);
    return out;
}

static PyMethodDef methods[] = {
    {"pack", tkm_pack, METH_O, "pack(texts) -> (UTF-8 of the texts back to back, len(texts) + 1 uint64 offsets as bytes)"},
    {"unpack", tkm_unpack, METH_VARARGS, "unpack(tokens uint32 buffer, tok_off uint64 buffer) -> list[list[int]]"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_tk_marshal", "str / list marshalling of the batch entry points", -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__tk_marshal(void) { return PyModule_Create(&moddef); }
