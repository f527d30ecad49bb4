// oracle build glue: replaces agg_nunique_string.cpp (needs hopscotch-fork internals; strings are out of scope)
#include <pybind11/pybind11.h>
namespace py = pybind11;
namespace vaex {
class Aggregator;
void add_agg_nunique_string(py::module &m, py::class_<Aggregator> &base) {}
} // namespace vaex
