/* fo_pyhost.c -- host-side helper of the drop-in boundary (CPython extension `_fo_pyhost`, built by __graft_entry__.build()).
 *
 * The reference's planner hands FOInterface one trajectory OBJECT per candidate (`trajectory.cartesian.{x,y,theta,v,a}`, five
 * numpy arrays of T samples: collision_probability.py:32,97; harm_model.py:81-94; be.py:90-94), and calls
 * trajectory_safety_assessment M times (interface.py:216-219).  The batched entry of this build needs the same data as five
 * [M][T] float64 blocks in ONE pinned staging buffer.  Gathering them in Python costs ~1.3 us per array (attribute lookups,
 * a list, np.concatenate): 3.6 ms for 2 000 objects, forty times the GPU step they feed.  Asking numpy for 10 000 data
 * pointers from Python is slower still (ndarray.ctypes.data: 2 us each).  So the walk over the objects happens here, with
 * the C API: two attribute lookups and one memcpy of T doubles per array, straight into the staging buffer.
 *
 *   pack_trajectories(objs: list, out: float64 ndarray [5][M][T], names: tuple of 5 str) -> None
 *
 * An array that is not a contiguous float64 vector (a list, float32, a strided view) is converted by numpy on the way; a
 * vector of another length raises ValueError like the Python path it replaces.  Not part of the C ABI of libfo_hip.so
 * (include/fo_hip.h): that one takes device pointers and is language neutral; this file is glue between CPython objects and
 * host memory, and the Python package falls back to its numpy statement of the same loop when the module is not built. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

static PyObject *s_cartesian;

static PyObject *pack_trajectories(PyObject *self, PyObject *args) {
  PyObject *objs, *out_o, *names;
  if (!PyArg_ParseTuple(args, "OOO", &objs, &out_o, &names)) return NULL;
  if (!PyList_Check(objs) && !PyTuple_Check(objs)) {
    PyErr_SetString(PyExc_TypeError, "pack_trajectories: objs must be a list or a tuple");
    return NULL;
  }
  if (!PyArray_Check(out_o)) {
    PyErr_SetString(PyExc_TypeError, "pack_trajectories: out must be a numpy array");
    return NULL;
  }
  PyArrayObject *out = (PyArrayObject *)out_o;
  if (PyArray_TYPE(out) != NPY_DOUBLE || PyArray_NDIM(out) != 3 || !PyArray_IS_C_CONTIGUOUS(out) || !PyArray_ISWRITEABLE(out)) {
    PyErr_SetString(PyExc_ValueError, "pack_trajectories: out must be a writeable C-contiguous float64 array [n][M][T]");
    return NULL;
  }
  if (!PyTuple_Check(names) || PyTuple_GET_SIZE(names) != PyArray_DIM(out, 0)) {
    PyErr_SetString(PyExc_ValueError, "pack_trajectories: one attribute name per block of out");
    return NULL;
  }
  const Py_ssize_t n = PyArray_DIM(out, 0), M = PyArray_DIM(out, 1), T = PyArray_DIM(out, 2);
  const Py_ssize_t len = PySequence_Fast_GET_SIZE(objs);
  if (len != M) {
    PyErr_Format(PyExc_ValueError, "pack_trajectories: %zd objects for %zd rows", len, M);
    return NULL;
  }
  double *base = (double *)PyArray_DATA(out);
  PyObject **items = PySequence_Fast_ITEMS(objs);
  for (Py_ssize_t m = 0; m < M; ++m) {
    PyObject *cart = PyObject_GetAttr(items[m], s_cartesian);
    if (!cart) return NULL;
    for (Py_ssize_t f = 0; f < n; ++f) {
      PyObject *arr = PyObject_GetAttr(cart, PyTuple_GET_ITEM(names, f));
      if (!arr) { Py_DECREF(cart); return NULL; }
      double *dst = base + ((size_t)f * (size_t)M + (size_t)m) * (size_t)T;
      if (PyArray_Check(arr) && PyArray_TYPE((PyArrayObject *)arr) == NPY_DOUBLE && PyArray_NDIM((PyArrayObject *)arr) == 1 &&
          PyArray_IS_C_CONTIGUOUS((PyArrayObject *)arr) && PyArray_ISALIGNED((PyArrayObject *)arr)) {
        if (PyArray_DIM((PyArrayObject *)arr, 0) != T) {
          Py_DECREF(arr); Py_DECREF(cart);
          PyErr_SetString(PyExc_ValueError, "all trajectories of a batch must have the same number of samples");
          return NULL;
        }
        memcpy(dst, PyArray_DATA((PyArrayObject *)arr), (size_t)T * sizeof(double));
      } else {   /* anything numpy can read as a vector of doubles */
        PyArrayObject *conv = (PyArrayObject *)PyArray_FROM_OTF(arr, NPY_DOUBLE, NPY_ARRAY_IN_ARRAY);
        if (!conv) { Py_DECREF(arr); Py_DECREF(cart); return NULL; }
        if (PyArray_SIZE(conv) != T) {
          Py_DECREF(conv); Py_DECREF(arr); Py_DECREF(cart);
          PyErr_SetString(PyExc_ValueError, "all trajectories of a batch must have the same number of samples");
          return NULL;
        }
        memcpy(dst, PyArray_DATA(conv), (size_t)T * sizeof(double));
        Py_DECREF(conv);
      }
      Py_DECREF(arr);
    }
    Py_DECREF(cart);
  }
  Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"pack_trajectories", pack_trajectories, METH_VARARGS,
     "pack_trajectories(objs, out[n][M][T] float64, names): out[f][m] = objs[m].cartesian.<names[f]>"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fo_pyhost", "host-side packing helper of frenetix_occlusion", -1, methods};

PyMODINIT_FUNC PyInit__fo_pyhost(void) {
  import_array();
  s_cartesian = PyUnicode_InternFromString("cartesian");
  if (!s_cartesian) return NULL;
  return PyModule_Create(&moduledef);
}
