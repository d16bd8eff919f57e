// CPython extension `scann._listwalk`: one-time conversion of the reference's nested-list dataset format to flat CSR
// arrays.  The reference keeps, per structure, a list over atoms of lists over neighbours of
// [species, idx, solid_angle, ratio, distance] records (voronoi_neighbor.py:38-47) and walks them in Python for every batch
// (datagenerator.py:69-135); here they are walked once, natively.
//
//   convert(data_neighbor, weight_index) -> (atoms_per_structure i64, degree_per_atom i64,
//                                            edge_local i32, edge_weight f32, edge_dist f32)   as bytearrays
// record[1] = neighbour index inside the structure, record[weight_index] = weight column (2 raw solid angle / 3
// normalised, datagenerator.py:48-50), record[-1] = distance.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <vector>

namespace {

struct Ref {  // owned reference released on scope exit
  PyObject* p;
  explicit Ref(PyObject* o) : p(o) {}
  ~Ref() { Py_XDECREF(p); }
  Ref(const Ref&) = delete;
  Ref& operator=(const Ref&) = delete;
};

bool as_double(PyObject* o, double* out) {
  const double v = PyFloat_AsDouble(o);
  if (v == -1.0 && PyErr_Occurred()) return false;
  *out = v;
  return true;
}

template <class T>
PyObject* to_bytearray(const std::vector<T>& v) {
  return PyByteArray_FromStringAndSize(reinterpret_cast<const char*>(v.data()), (Py_ssize_t)(v.size() * sizeof(T)));
}

PyObject* convert(PyObject*, PyObject* args) {
  PyObject* data;
  long wi;
  if (!PyArg_ParseTuple(args, "Ol", &data, &wi)) return nullptr;
  Ref structs(PySequence_Fast(data, "data_neighbor must be a sequence of structures"));
  if (!structs.p) return nullptr;
  const Py_ssize_t ns = PySequence_Fast_GET_SIZE(structs.p);
  std::vector<int64_t> n_atoms, degree;
  std::vector<int32_t> local;
  std::vector<float> weight, dist;
  n_atoms.reserve(ns);
  for (Py_ssize_t s = 0; s < ns; ++s) {
    Ref atoms(PySequence_Fast(PySequence_Fast_GET_ITEM(structs.p, s), "a structure must be a sequence of atoms"));
    if (!atoms.p) return nullptr;
    const Py_ssize_t na = PySequence_Fast_GET_SIZE(atoms.p);
    n_atoms.push_back(na);
    for (Py_ssize_t a = 0; a < na; ++a) {
      Ref nbrs(PySequence_Fast(PySequence_Fast_GET_ITEM(atoms.p, a), "an atom must hold a sequence of neighbour records"));
      if (!nbrs.p) return nullptr;
      const Py_ssize_t nn = PySequence_Fast_GET_SIZE(nbrs.p);
      degree.push_back(nn);
      for (Py_ssize_t k = 0; k < nn; ++k) {
        Ref rec(PySequence_Fast(PySequence_Fast_GET_ITEM(nbrs.p, k), "a neighbour record must be a sequence"));
        if (!rec.p) return nullptr;
        const Py_ssize_t len = PySequence_Fast_GET_SIZE(rec.p);
        if (len < 2 || wi < 0 || wi >= len) {
          PyErr_Format(PyExc_ValueError, "neighbour record of length %zd has no column %ld (structure %zd, atom %zd)", len,
                       wi, s, a);
          return nullptr;
        }
        double idx, w, d;
        if (!as_double(PySequence_Fast_GET_ITEM(rec.p, 1), &idx) || !as_double(PySequence_Fast_GET_ITEM(rec.p, wi), &w) ||
            !as_double(PySequence_Fast_GET_ITEM(rec.p, len - 1), &d))
          return nullptr;
        local.push_back((int32_t)idx);
        weight.push_back((float)w);
        dist.push_back((float)d);
      }
    }
  }
  Ref r0(to_bytearray(n_atoms)), r1(to_bytearray(degree)), r2(to_bytearray(local)), r3(to_bytearray(weight)),
      r4(to_bytearray(dist));
  if (!r0.p || !r1.p || !r2.p || !r3.p || !r4.p) return nullptr;
  return PyTuple_Pack(5, r0.p, r1.p, r2.p, r3.p, r4.p);
}

PyMethodDef methods[] = {{"convert", convert, METH_VARARGS, "nested neighbour lists -> flat CSR pieces"},
                         {nullptr, nullptr, 0, nullptr}};
PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_listwalk", "native walker for the reference's nested-list datasets", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__listwalk(void) { return PyModule_Create(&moddef); }
