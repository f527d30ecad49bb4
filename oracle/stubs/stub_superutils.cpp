// oracle build glue: replaces hash_string.cpp / hash_object.cpp (strings & PyObject keys are out of scope);
// the empty classes only satisfy the name lookups in vaex/hash.py.
#include <pybind11/pybind11.h>
namespace py = pybind11;
namespace vaex {
struct D1 {}; struct D2 {}; struct D3 {}; struct D4 {};
void init_hash_string(py::module &m) {
    py::class_<D1>(m, "ordered_set_string");
    py::class_<D2>(m, "counter_string");
    py::class_<D3>(m, "index_hash_string");
    py::class_<D4>(m, "hash_map_string");
}
void init_hash_object(py::module &m) {}
} // namespace vaex
