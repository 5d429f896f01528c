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
#include <math.h>
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

/* obstacle_rows(P [O][Lmax][3] f64, L [O] i64, t0 [O] i64, static [O] u8, vx [O][4] f64, vy [O][4] f64, dims [O][2] f64,
 *               flags [O] u8, timestep: int, host [O*105] u8, present [O] u8) -> None
 * The obstacle rows of one time step in the layout SensorModel.stage_obstacles uploads (corners [O][4][2] | centres [O][2] |
 * headings [O] | dimensions [O][2] | flags [O]) -- utils/fo_obstacle.py FOObstacles.update, whose numpy statement of the
 * same arithmetic (fo_obstacle.py:79-116 + helper_functions.py:99-112: rel = step - initial, 0 -> initial state, >= 1 ->
 * state_list[rel-1], else absent; rectangle vertices rotated and shifted) costs ~40 us of interpreter time whatever O is.
 * Same operations in the same order, libm cos / sin like math.cos / math.sin (built with -ffp-contract=off): same bits. */
static int want(PyObject *o, int type, int ndim, const char *what) {
  if (!PyArray_Check(o) || PyArray_TYPE((PyArrayObject *)o) != type || PyArray_NDIM((PyArrayObject *)o) != ndim ||
      !PyArray_IS_C_CONTIGUOUS((PyArrayObject *)o)) {
    PyErr_Format(PyExc_ValueError, "obstacle_rows: %s must be a C-contiguous array of the documented type and rank", what);
    return 0;
  }
  return 1;
}
static PyObject *obstacle_rows(PyObject *self, PyObject *args) {
  PyObject *P_o, *L_o, *t0_o, *st_o, *vx_o, *vy_o, *dims_o, *fl_o, *host_o, *pres_o;
  long long timestep;
  if (!PyArg_ParseTuple(args, "OOOOOOOOLOO", &P_o, &L_o, &t0_o, &st_o, &vx_o, &vy_o, &dims_o, &fl_o, &timestep, &host_o, &pres_o))
    return NULL;
  if (!want(P_o, NPY_DOUBLE, 3, "P") || !want(L_o, NPY_INT64, 1, "L") || !want(t0_o, NPY_INT64, 1, "t0") ||
      !want(st_o, NPY_UINT8, 1, "static") || !want(vx_o, NPY_DOUBLE, 2, "vx") || !want(vy_o, NPY_DOUBLE, 2, "vy") ||
      !want(dims_o, NPY_DOUBLE, 2, "dims") || !want(fl_o, NPY_UINT8, 1, "flags") || !want(host_o, NPY_UINT8, 1, "host") ||
      !want(pres_o, NPY_UINT8, 1, "present"))
    return NULL;
  PyArrayObject *Pa = (PyArrayObject *)P_o;
  const npy_intp O = PyArray_DIM(Pa, 0), Lmax = PyArray_DIM(Pa, 1);
  if (PyArray_DIM(Pa, 2) != 3 || PyArray_DIM((PyArrayObject *)L_o, 0) != O || PyArray_DIM((PyArrayObject *)t0_o, 0) != O ||
      PyArray_DIM((PyArrayObject *)st_o, 0) != O || PyArray_DIM((PyArrayObject *)vx_o, 0) != O || PyArray_DIM((PyArrayObject *)vx_o, 1) != 4 ||
      PyArray_DIM((PyArrayObject *)vy_o, 0) != O || PyArray_DIM((PyArrayObject *)vy_o, 1) != 4 || PyArray_DIM((PyArrayObject *)dims_o, 0) != O ||
      PyArray_DIM((PyArrayObject *)dims_o, 1) != 2 || PyArray_DIM((PyArrayObject *)fl_o, 0) != O ||
      PyArray_DIM((PyArrayObject *)host_o, 0) != O * 105 || PyArray_DIM((PyArrayObject *)pres_o, 0) != O ||
      !PyArray_ISWRITEABLE((PyArrayObject *)host_o) || !PyArray_ISWRITEABLE((PyArrayObject *)pres_o)) {
    PyErr_SetString(PyExc_ValueError, "obstacle_rows: array shapes do not fit together");
    return NULL;
  }
  const double *P = (const double *)PyArray_DATA(Pa), *vx = (const double *)PyArray_DATA((PyArrayObject *)vx_o);
  const double *vy = (const double *)PyArray_DATA((PyArrayObject *)vy_o), *dims = (const double *)PyArray_DATA((PyArrayObject *)dims_o);
  const npy_int64 *L = (const npy_int64 *)PyArray_DATA((PyArrayObject *)L_o), *t0 = (const npy_int64 *)PyArray_DATA((PyArrayObject *)t0_o);
  const npy_uint8 *st = (const npy_uint8 *)PyArray_DATA((PyArrayObject *)st_o), *fl = (const npy_uint8 *)PyArray_DATA((PyArrayObject *)fl_o);
  char *host = (char *)PyArray_DATA((PyArrayObject *)host_o);
  npy_uint8 *pres = (npy_uint8 *)PyArray_DATA((PyArrayObject *)pres_o);
  double *corn = (double *)host, *cen = (double *)(host + O * 64), *yaw = (double *)(host + O * 80), *dm = (double *)(host + O * 88);
  npy_uint8 *flags = (npy_uint8 *)(host + O * 104);
  memset(host, 0, (size_t)O * 105);
  for (npy_intp i = 0; i < O; ++i) {
    const long long rel = st[i] ? 0 : timestep - t0[i];
    pres[i] = rel >= 0 && rel < L[i];
    if (!pres[i]) continue;
    const double *q = P + ((size_t)i * Lmax + (size_t)rel) * 3;
    const double c = cos(q[2]), s = sin(q[2]);
    for (int k = 0; k < 4; ++k) {
      const double ax = vx[i * 4 + k], ay = vy[i * 4 + k];
      const double cx = c * ax, sy = s * ay, sx = s * ax, cy = c * ay;
      corn[(i * 4 + k) * 2 + 0] = q[0] + (cx - sy);
      corn[(i * 4 + k) * 2 + 1] = q[1] + (sx + cy);
    }
    cen[i * 2] = q[0]; cen[i * 2 + 1] = q[1];
    yaw[i] = q[2];
    dm[i * 2] = dims[i * 2]; dm[i * 2 + 1] = dims[i * 2 + 1];
    flags[i] = fl[i];
  }
  Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"pack_trajectories", pack_trajectories, METH_VARARGS,
     "pack_trajectories(objs, out[n][M][T] float64, names): out[f][m] = objs[m].cartesian.<names[f]>"},
    {"obstacle_rows", obstacle_rows, METH_VARARGS,
     "obstacle_rows(P, L, t0, static, vx, vy, dims, flags, timestep, host, present): the obstacle rows of one time step"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fo_pyhost", "host-side packing helper of frenetix_occlusion", -1, methods};

PyMODINIT_FUNC PyInit__fo_pyhost(void) {
  import_array();
  s_cartesian = PyUnicode_InternFromString("cartesian");
  if (!s_cartesian) return NULL;
  return PyModule_Create(&moduledef);
}
