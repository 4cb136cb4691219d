/* maro_amd._fastobj — the per-env Python objects of the reference-shaped OBJECT API (GpuVectorEnv.step), built in C.
 *
 * A whole-batch step hands out one DecisionEvent and one metrics dict per env and takes one Action per env
 * (maro/vector_env/vector_env.py:116-144, maro/simulator/scenarios/cim/common.py:25-150).  Creating and reading those objects
 * is what an object-API step costs once the engine is on the GPU; the three loops below do it without the interpreter's
 * per-bytecode overhead (~0.1-0.25 us per object instead of ~0.6-1.0).  Host-side glue only: no device code, no simulation
 * logic — maro_amd/cim/vector_env.py falls back to its own comprehensions when this module is not built.
 *
 *   encode_actions(actions, skip, out_acts, out_nact, A, load_obj, discharge_obj) -> list of (env, entry) the caller must encode itself
 *   build_events(cls, rows, stride, want, snaps, scope_cls)                       -> list of DecisionEvent | None (action scope objects included)
 *   build_metrics(met, live, k0, k1, k2)                                          -> list of dict | None
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *s_vessel_idx, *s_port_idx, *s_quantity, *s_action_type, *s_name;
static PyObject *k_tick, *k_port, *k_vessel, *k_snap, *k_scope_obj, *k_early, *k_scope_fn, *k_early_fn, *k_scope, *k_load, *k_discharge;

static int as_long(PyObject* o, long* out) {
  long v = PyLong_AsLong(o);
  if (v == -1 && PyErr_Occurred()) {
    PyErr_Clear();
    PyObject* i = PyNumber_Index(o);      /* numpy integers and friends */
    if (!i) return -1;
    v = PyLong_AsLong(i);
    Py_DECREF(i);
    if (v == -1 && PyErr_Occurred()) return -1;
  }
  *out = v;
  return 0;
}

static PyObject* encode_actions(PyObject* self, PyObject* args) {
  PyObject *actions, *load_obj, *dis_obj;
  Py_buffer skip, acts, nact;
  int A;
  if (!PyArg_ParseTuple(args, "Oy*w*w*iOO", &actions, &skip, &acts, &nact, &A, &load_obj, &dis_obj)) return NULL;
  PyObject* multi = PyList_New(0);
  PyObject* seq = PySequence_Fast(actions, "actions must be a list");
  if (!seq || !multi) goto fail;
  {
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
    if ((Py_ssize_t)skip.len < n || (Py_ssize_t)nact.len < n * 4 || (Py_ssize_t)acts.len < n * A * 16) {
      PyErr_SetString(PyExc_ValueError, "encode_actions: buffers smaller than the action list");
      goto fail;
    }
    const uint8_t* sk = (const uint8_t*)skip.buf;
    int32_t* a = (int32_t*)acts.buf;
    int32_t* na = (int32_t*)nact.buf;
    PyObject** items = PySequence_Fast_ITEMS(seq);
    for (Py_ssize_t e = 0; e < n; e++) {
      PyObject* x = items[e];
      if (x == Py_None || sk[e]) continue;
      if (PyList_Check(x) || PyTuple_Check(x)) {
        PyObject* pair = Py_BuildValue("(nO)", e, x);
        if (!pair || PyList_Append(multi, pair) < 0) { Py_XDECREF(pair); goto fail; }
        Py_DECREF(pair);
        continue;
      }
      long v[4];
      PyObject* names[3] = {s_vessel_idx, s_port_idx, s_quantity};
      for (int j = 0; j < 3; j++) {
        PyObject* o = PyObject_GetAttr(x, names[j]);
        if (!o) goto fail;
        const int rc = as_long(o, &v[j]);
        Py_DECREF(o);
        if (rc < 0) goto fail;
      }
      PyObject* t = PyObject_GetAttr(x, s_action_type);
      if (!t) goto fail;
      if (t == load_obj) v[3] = 0;
      else if (t == dis_obj) v[3] = 1;
      else {   /* a foreign enum with the same member names (payloads.action_code) */
        PyObject* nm = PyObject_GetAttr(t, s_name);
        if (!nm) { PyErr_Clear(); nm = PyObject_Str(t); }
        if (!nm) { Py_DECREF(t); goto fail; }
        PyObject* up = PyObject_CallMethod(nm, "upper", NULL);
        Py_DECREF(nm);
        if (!up) { Py_DECREF(t); goto fail; }
        const char* c = PyUnicode_AsUTF8(up);
        const size_t L = c ? strlen(c) : 0;
        v[3] = (L >= 9 && strcmp(c + L - 9, "DISCHARGE") == 0) ? 1 : 0;
        Py_DECREF(up);
      }
      Py_DECREF(t);
      int32_t* row = a + (size_t)e * A * 4;
      row[0] = (int32_t)v[0]; row[1] = (int32_t)v[1]; row[2] = (int32_t)v[2]; row[3] = (int32_t)v[3];
      na[e] = 1;
    }
  }
  Py_DECREF(seq);
  PyBuffer_Release(&skip); PyBuffer_Release(&acts); PyBuffer_Release(&nact);
  return multi;
fail:
  Py_XDECREF(seq);
  Py_XDECREF(multi);
  PyBuffer_Release(&skip); PyBuffer_Release(&acts); PyBuffer_Release(&nact);
  return NULL;
}

/* rows: int32 [n][stride] C-contiguous (tick, port, vessel, load, discharge, early, ...); want: uint8 [n]; snaps: list of n objects */
static PyObject* build_events(PyObject* self, PyObject* args) {
  PyObject *cls, *snaps, *scope_cls;
  Py_buffer rows, want;
  int stride;
  if (!PyArg_ParseTuple(args, "Oy*iy*OO", &cls, &rows, &stride, &want, &snaps, &scope_cls)) return NULL;
  PyObject* out = NULL;
  if (!PyType_Check(cls) || !PyList_Check(snaps) || !PyType_Check(scope_cls)) { PyErr_SetString(PyExc_TypeError, "build_events(cls, rows, stride, want, snaps, scope_cls)"); goto fail; }
  {
    PyTypeObject* tp = (PyTypeObject*)cls;
    PyTypeObject* stp = (PyTypeObject*)scope_cls;
    const Py_ssize_t n = PyList_GET_SIZE(snaps);
    if ((Py_ssize_t)want.len < n || (Py_ssize_t)rows.len < n * stride * 4 || stride < 6) { PyErr_SetString(PyExc_ValueError, "build_events: buffers smaller than the batch"); goto fail; }
    const int32_t* r = (const int32_t*)rows.buf;
    const uint8_t* w = (const uint8_t*)want.buf;
    out = PyList_New(n);
    if (!out) goto fail;
    for (Py_ssize_t e = 0; e < n; e++, r += stride) {
      if (!w[e]) { Py_INCREF(Py_None); PyList_SET_ITEM(out, e, Py_None); continue; }
      PyObject* ev = tp->tp_alloc(tp, 0);
      if (!ev) goto fail;
      PyList_SET_ITEM(out, e, ev);
      PyObject** dp = _PyObject_GetDictPtr(ev);
      if (!dp) { PyErr_SetString(PyExc_TypeError, "DecisionEvent class without an instance dict"); goto fail; }
      PyObject* d = _PyDict_NewPresized(6);   /* six instance attributes: the class holds the defaults of the rest (payloads.py) */
      if (!d) goto fail;
      Py_XSETREF(*dp, d);
      PyObject *t = PyLong_FromLong(r[0]), *p = PyLong_FromLong(r[1]), *v = PyLong_FromLong(r[2]), *ed = PyLong_FromLong(r[5]);
      PyObject *ld = PyLong_FromLong(r[3]), *dc = PyLong_FromLong(r[4]);
      /* the ActionScope(load, discharge) object itself (cim/common.py:56-69: two plain attributes), so that reading event.action_scope
         costs the agent an attribute lookup instead of a Python-level constructor call */
      PyObject* sc = stp->tp_alloc(stp, 0);
      int bad = !t || !p || !v || !ed || !ld || !dc || !sc;
      if (!bad) {
        PyObject** sdp = _PyObject_GetDictPtr(sc);
        PyObject* sd = sdp ? _PyDict_NewPresized(2) : NULL;
        bad = !sd;
        if (!bad) {
          Py_XSETREF(*sdp, sd);
          bad = PyDict_SetItem(sd, k_load, ld) < 0 || PyDict_SetItem(sd, k_discharge, dc) < 0;
        }
      }
      if (!bad) {
        bad |= PyDict_SetItem(d, k_tick, t) < 0 || PyDict_SetItem(d, k_port, p) < 0 || PyDict_SetItem(d, k_vessel, v) < 0;
        bad |= PyDict_SetItem(d, k_snap, PyList_GET_ITEM(snaps, e)) < 0 || PyDict_SetItem(d, k_early, ed) < 0 || PyDict_SetItem(d, k_scope_obj, sc) < 0;
      }
      Py_XDECREF(t); Py_XDECREF(p); Py_XDECREF(v); Py_XDECREF(ed); Py_XDECREF(ld); Py_XDECREF(dc); Py_XDECREF(sc);
      if (bad) goto fail;
    }
  }
  PyBuffer_Release(&rows); PyBuffer_Release(&want);
  return out;
fail:
  Py_XDECREF(out);
  PyBuffer_Release(&rows); PyBuffer_Release(&want);
  return NULL;
}

/* met: int64 [n][3]; live: uint8 [n] */
static PyObject* build_metrics(PyObject* self, PyObject* args) {
  Py_buffer met, live;
  PyObject *k0, *k1, *k2;
  if (!PyArg_ParseTuple(args, "y*y*OOO", &met, &live, &k0, &k1, &k2)) return NULL;
  const Py_ssize_t n = live.len;
  PyObject* out = NULL;
  if ((Py_ssize_t)met.len < n * 24) { PyErr_SetString(PyExc_ValueError, "build_metrics: metrics buffer smaller than the batch"); goto fail; }
  out = PyList_New(n);
  if (!out) goto fail;
  {
    const int64_t* m = (const int64_t*)met.buf;
    const uint8_t* lv = (const uint8_t*)live.buf;
    for (Py_ssize_t e = 0; e < n; e++, m += 3) {
      if (!lv[e]) { Py_INCREF(Py_None); PyList_SET_ITEM(out, e, Py_None); continue; }
      PyObject* d = _PyDict_NewPresized(3);
      if (!d) goto fail;
      PyList_SET_ITEM(out, e, d);
      PyObject *a = PyLong_FromLongLong(m[0]), *b = PyLong_FromLongLong(m[1]), *c = PyLong_FromLongLong(m[2]);
      int bad = !a || !b || !c;
      if (!bad) bad = PyDict_SetItem(d, k0, a) < 0 || PyDict_SetItem(d, k1, b) < 0 || PyDict_SetItem(d, k2, c) < 0;
      Py_XDECREF(a); Py_XDECREF(b); Py_XDECREF(c);
      if (bad) goto fail;
    }
  }
  PyBuffer_Release(&met); PyBuffer_Release(&live);
  return out;
fail:
  Py_XDECREF(out);
  PyBuffer_Release(&met); PyBuffer_Release(&live);
  return NULL;
}

static PyMethodDef methods[] = {
  {"encode_actions", encode_actions, METH_VARARGS, "list of Action | None | list -> int32 action rows (in place); returns the (env, entry) pairs left to the caller"},
  {"build_events", build_events, METH_VARARGS, "decision rows -> list of DecisionEvent | None"},
  {"build_metrics", build_metrics, METH_VARARGS, "metrics rows -> list of dict | None"},
  {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_fastobj", "C loops behind GpuVectorEnv's whole-batch step (object API)", -1, methods};

PyMODINIT_FUNC PyInit__fastobj(void) {
#define S(var, text) if (!(var = PyUnicode_InternFromString(text))) return NULL
  S(s_vessel_idx, "vessel_idx"); S(s_port_idx, "port_idx"); S(s_quantity, "quantity"); S(s_action_type, "action_type"); S(s_name, "name");
  S(k_tick, "tick"); S(k_port, "port_idx"); S(k_vessel, "vessel_idx"); S(k_snap, "snapshot_list"); S(k_scope_obj, "_action_scope");
  S(k_early, "_early_discharge"); S(k_scope_fn, "_action_scope_func"); S(k_early_fn, "_early_discharge_func"); S(k_scope, "_scope");
  S(k_load, "load"); S(k_discharge, "discharge");
#undef S
  return PyModule_Create(&moddef);
}
