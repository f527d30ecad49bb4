// pybind11 shim: re-exposes the C-ABI of libvaexhip.so (include/vaex_hip.h) as the Python class
// surface of the reference's native module `vaex.superagg`
// (/root/reference/packages/vaex-core/src/agg.cpp:91-118, binners.cpp:92-146,
// binner_ordinal.cpp:205-253, agg_base.hpp:249-261, agg_sum.cpp:213-230), so vaex's unmodified
// Python (vaex/cpu.py:44-65, :630-845; vaex/agg.py:278-335) can drive the HIP kernels:
//
//     Grid, Binner, Aggregator,
//     BinnerScalar_<T>[_non_native], BinnerOrdinal_<T>[_non_native], BinnerHash_<T>[_non_native],
//     AggCount_<T>, AggSum_<T>, AggSumMoment_<T>, AggMin_<T>, AggMax_<T>   [_non_native]
//     ordered_set_<T>  (the subset of vaex.superutils the groupby path uses)
//
// for T in float64 float32 int64 int32 int16 int8 uint64 uint32 uint16 uint8 bool.
// No computation happens here: every method forwards to one extern "C" entry point.  Extension over
// the reference: set_data / set_data_mask also accept objects exposing __cuda_array_interface__
// (e.g. torch tensors on the GPU) — HBM-resident columns are then used in place.
#include <pybind11/numpy.h>
#include <atomic>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "vaex_hip.h"

namespace py = pybind11;

namespace {

const char *kTypeNames[VXH_DTYPE_COUNT] = {"float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"};
const int kTypeSizes[VXH_DTYPE_COUNT] = {8, 4, 8, 4, 2, 1, 8, 4, 2, 1, 1};
const char *kNumpyFormats[VXH_DTYPE_COUNT] = {"d", "f", "q", "i", "h", "b", "Q", "I", "H", "B", "?"};

void check(int rc) {
    if (rc != 0) throw std::runtime_error(vxh_last_error());
}

// a borrowed 1-d array: host buffer or device array
struct ArrayRef {
    const void *ptr = nullptr;
    uint64_t n = 0;
    int mem = VXH_MEM_HOST;
    ssize_t itemsize = 0;
};

ArrayRef resolve_array(const py::object &obj) {
    ArrayRef r;
    if (py::hasattr(obj, "__cuda_array_interface__")) {
        py::dict cai = obj.attr("__cuda_array_interface__");
        py::tuple shape = cai["shape"];
        if (shape.size() != 1) throw std::runtime_error("Expected a 1d array");
        py::tuple data = cai["data"];
        std::string typestr = py::str(cai["typestr"]);
        r.itemsize = std::stoi(typestr.substr(2));
        r.n = shape[0].cast<uint64_t>();
        if (cai.contains("strides") && !cai["strides"].is_none()) {
            py::tuple strides = cai["strides"];
            if (strides.size() != 1 || (r.n > 1 && strides[0].cast<ssize_t>() != r.itemsize)) throw std::runtime_error("Expected a contiguous array");
        }
        r.ptr = reinterpret_cast<const void *>(data[0].cast<uintptr_t>());
        r.mem = VXH_MEM_DEVICE;
        return r;
    }
    py::buffer buf = py::reinterpret_borrow<py::buffer>(obj);
    py::buffer_info info = buf.request();
    if (info.ndim != 1) throw std::runtime_error("Expected a 1d array");
    if (info.shape[0] > 1 && info.strides[0] != info.itemsize) throw std::runtime_error("Expected a contiguous array");
    r.ptr = info.ptr;
    r.n = (uint64_t)info.shape[0];
    r.itemsize = info.itemsize;
    r.mem = VXH_MEM_HOST;
    return r;
}

// a mask handed over next to `data` must live in the same memory space (a host pointer is not dereferenceable by a
// kernel reading device data, and vice versa) and cover every row
void check_mask(const ArrayRef &data, const ArrayRef &mask) {
    if (mask.mem != data.mem) throw std::runtime_error("mask must live where the data lives (host with host, device with device)");
    if (mask.n < data.n) throw std::runtime_error("mask is shorter than the data");
    if (mask.itemsize != 1) throw std::runtime_error("mask must be a 1-byte (bool / uint8) array");
}

// ------------------------------------------------------------------------------------------
// lifetime of device arrays (include/vaex_hip.h "Data pointers": VXH_MEM_DEVICE)
// ------------------------------------------------------------------------------------------
// Grid.bin over device columns returns with its kernels ENQUEUED on the slot's stream: the columns are read after the call has
// returned.  The reference's contract — the caller keeps its arrays alive for the duration of bin() (vaex/cpu.py:708-710) — is all a
// Python caller knows, so the bookkeeping is done here: every object that was handed a device array keeps a reference to it per thread
// slot; a reference that is replaced (the next chunk's set_data), cleared or orphaned by its holder's destruction is RETIRED to its
// slot's list and dropped only once that slot is idle (vxh_slot_busy).  A torch tensor freed by the caller right after bin() therefore
// does not go back to torch's caching allocator — which would hand the block to the next torch.empty on ANOTHER stream — while a
// kernel of this library still reads it.  (All of this runs under the GIL; the lists are never destroyed: no decref at interpreter exit.)
std::map<int, std::vector<py::object>> &retired_refs() {
    static auto *m = new std::map<int, std::vector<py::object>>();
    return *m;
}
constexpr size_t kMaxRetiredPerSlot = 128; // (beyond this the slot is waited for: the list must not grow with the number of chunks)

void collect_retired(int thread) {
    auto &all = retired_refs();
    auto it = all.find(thread);
    if (it == all.end() || it->second.empty()) return;
    int busy = 1;
    if (vxh_slot_busy(thread, &busy) != 0) return; // (a failing query keeps the references: leaking is the safe side)
    if (busy && it->second.size() >= kMaxRetiredPerSlot) {
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_slot_wait(thread);
        }
        busy = rc != 0;
        it = all.find(thread); // (the GIL was released: another thread may have touched the map)
        if (it == all.end()) return;
    }
    if (!busy) {
        std::vector<py::object> drop;
        drop.swap(it->second); // (decrefs may run arbitrary Python: after the list is empty)
    }
}

void retire_ref(int thread, py::object &&o) {
    if (!o || o.is_none()) return;
    int busy = 1;
    if (vxh_slot_busy(thread, &busy) == 0 && !busy) return; // (nothing in flight on that slot: dropped here)
    retired_refs()[thread].push_back(std::move(o));
}

void collect_all_retired() { // after vxh_synchronize: nothing is in flight anywhere
    std::vector<py::object> drop;
    for (auto &kv : retired_refs()) {
        for (auto &o : kv.second) drop.push_back(std::move(o));
        kv.second.clear();
    }
}

// the device arrays one binner / aggregator / selection currently points at: (thread slot, index) -> reference
struct HeldArrays {
    std::map<std::pair<int, int>, py::object> refs;
    void hold(int thread, int index, const py::object &obj, const ArrayRef &a) {
        collect_retired(thread);
        release(thread, index);
        if (a.mem == VXH_MEM_DEVICE) refs[{thread, index}] = obj;
    }
    void release(int thread, int index) {
        auto it = refs.find({thread, index});
        if (it == refs.end()) return;
        py::object o = std::move(it->second);
        refs.erase(it);
        retire_ref(thread, std::move(o));
    }
    ~HeldArrays() {
        for (auto &kv : refs) retire_ref(kv.first.first, std::move(kv.second));
    }
};

// ------------------------------------------------------------------------------------------
// hash map
// ------------------------------------------------------------------------------------------
struct PyHashMap {
    vxh_hashmap *h = nullptr;
    int dtype;
    PyHashMap(int dtype, uint64_t hint) : dtype(dtype) { check(vxh_hashmap_create(dtype, hint, &h)); }
    virtual ~PyHashMap() { vxh_hashmap_destroy(h); }
    PyHashMap(const PyHashMap &) = delete;

    void update(const py::object &keys, const py::object &mask) {
        ArrayRef k = resolve_array(keys);
        if (k.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and hash map are not equal");
        const uint8_t *mp = nullptr;
        if (!mask.is_none()) {
            ArrayRef m = resolve_array(mask);
            if (m.mem != k.mem || m.n < k.n) throw std::runtime_error("mask must live where the keys live and be as long");
            mp = (const uint8_t *)m.ptr;
        }
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_hashmap_update(h, k.ptr, mp, k.n, k.mem);
        }
        check(rc);
    }
    void set_keys(py::array_t<int64_t, py::array::c_style | py::array::forcecast> keys) {
        check(vxh_hashmap_set_keys(h, keys.data(), (uint64_t)keys.size()));
    }
    int64_t count() {
        int64_t c;
        check(vxh_hashmap_count(h, &c));
        return c;
    }
    int64_t null_index() {
        int64_t c;
        check(vxh_hashmap_null_index(h, &c));
        return c;
    }
    py::array_t<int64_t> map_ordinal(const py::object &keys) {
        ArrayRef k = resolve_array(keys);
        if (k.mem != VXH_MEM_HOST) throw std::runtime_error("map_ordinal: host arrays only (use BinnerHash for device-resident keys)");
        if (k.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and hash map are not equal");
        py::array_t<int64_t> out((ssize_t)k.n);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_hashmap_map_ordinal(h, k.ptr, k.n, VXH_MEM_HOST, out.mutable_data());
        }
        check(rc);
        return out;
    }
    // device-resident keys -> device-resident ordinals (one lookup kernel; -1 = unknown key)
    void map_ordinal_device(const py::object &keys, uintptr_t out_ptr, uint64_t out_n) {
        ArrayRef k = resolve_array(keys);
        if (k.mem != VXH_MEM_DEVICE) throw std::runtime_error("map_ordinal_device: device arrays only");
        if (k.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and hash map are not equal");
        if (out_n < k.n) throw std::runtime_error("map_ordinal_device: output too short");
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_hashmap_map_ordinal(h, k.ptr, k.n, VXH_MEM_DEVICE, (int64_t *)out_ptr);
        }
        check(rc);
    }
    py::array_t<int64_t> key_array() {
        py::array_t<int64_t> out((ssize_t)count());
        check(vxh_hashmap_keys(h, out.mutable_data()));
        return out;
    }
};
template <int DT>
struct THashMap : PyHashMap {
    explicit THashMap(uint64_t hint) : PyHashMap(DT, hint) {}
};

// ------------------------------------------------------------------------------------------
// results of the row-wise helpers: a device array any consumer of __cuda_array_interface__ (set_data, torch) can read
// ------------------------------------------------------------------------------------------
struct PyDeviceArray {
    void *p = nullptr;
    uint64_t n = 0;
    int dtype = VXH_F64;
    PyDeviceArray(uint64_t n, int dtype) : n(n), dtype(dtype) { check(vxh_device_alloc((size_t)n * kTypeSizes[dtype], &p)); }
    ~PyDeviceArray() { vxh_device_free(p); }
    PyDeviceArray(const PyDeviceArray &) = delete;
    py::dict interface() const {
        static const char *typestr[VXH_DTYPE_COUNT] = {"<f8", "<f4", "<i8", "<i4", "<i2", "|i1", "<u8", "<u4", "<u2", "|u1", "|b1"};
        py::dict d;
        d["shape"] = py::make_tuple(n);
        d["typestr"] = typestr[dtype];
        d["data"] = py::make_tuple((uintptr_t)p, false);
        d["version"] = 3;
        d["strides"] = py::none();
        return d;
    }
};

// ------------------------------------------------------------------------------------------
// hash groupby in one partitioned pass (vxh_groupby_*)
// ------------------------------------------------------------------------------------------
py::array pinned_array(uint64_t elems, const std::string &fmt) { // 8-byte elements in page-locked memory owned by the array
    void *p = nullptr;
    check(vxh_host_alloc((size_t)std::max<uint64_t>(elems, 1) * 8, &p));
    py::capsule owner(p, [](void *q) { vxh_host_free(q); });
    return py::array(py::dtype(fmt), std::vector<ssize_t>{(ssize_t)elems}, std::vector<ssize_t>{8}, p, owner);
}

struct PyGroupBy {
    vxh_groupby *h = nullptr;
    int nv = 1;
    ~PyGroupBy() { vxh_groupby_destroy(h); }
    PyGroupBy() = default;
    PyGroupBy(const PyGroupBy &) = delete;
    uint64_t size() const { return vxh_groupby_size(h); }
    py::array column(int which, int value_index) {
        const bool integer = which == VXH_GB_KEYS || which == VXH_GB_ROWS || which == VXH_GB_COUNT;
        py::array out = pinned_array(size(), integer ? "q" : "d");
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_groupby_column(h, value_index, which, out.mutable_data());
        }
        check(rc);
        return out;
    }
    py::dict info() const {
        static const char *names[] = {"buckets", "slots", "retries", "ms_scatter", "ms_reduce", "ms_sort", "compact_records", "heavy_keys_in_pass", "direct_table", "tag_table"};
        py::dict d;
        for (int i = 0; i < 10; i++) {
            double v = 0;
            check(vxh_groupby_info(h, i, &v));
            d[names[i]] = v;
        }
        return d;
    }
};

// ------------------------------------------------------------------------------------------
// binners
// ------------------------------------------------------------------------------------------
struct PyBinner {
    vxh_binner *h = nullptr;
    int dtype = 0;
    bool flip = false;
    int threads = 1;
    std::string expression;
    HeldArrays held; // index 0: data, 1: mask
    virtual ~PyBinner() { vxh_binner_destroy(h); }
    PyBinner() = default;
    PyBinner(const PyBinner &o) : dtype(o.dtype), flip(o.flip), threads(o.threads), expression(o.expression) {
        check(vxh_binner_copy(o.h, &h));
        held.refs = o.held.refs; // (whatever pointers the copy inherited stay alive with it)
    }

    void set_data(int thread, const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        if (a.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and binner are not equal");
        check(vxh_binner_set_data(h, thread, a.ptr, a.n, a.mem));
        held.hold(thread, 0, ar, a);
    }
    void set_data_mask(int thread, const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        check(vxh_binner_set_data_mask(h, thread, (const uint8_t *)a.ptr, a.n, a.mem));
        held.hold(thread, 1, ar, a);
    }
    void clear_data_mask(int thread) {
        check(vxh_binner_clear_data_mask(h, thread));
        held.release(thread, 1);
    }
    uint64_t shape() const { return vxh_binner_shape(h); }
};

struct PyBinnerScalar : PyBinner {
    double vmin, vmax;
    uint64_t bins;
    PyBinnerScalar(int dt, bool fl, int threads_, std::string expr, double vmin, double vmax, uint64_t bins) : vmin(vmin), vmax(vmax), bins(bins) {
        dtype = dt; flip = fl; threads = threads_; expression = std::move(expr);
        check(vxh_binner_scalar_create(threads, dt, fl, vmin, vmax, bins, &h));
    }
};
template <int DT, bool FLIP>
struct TBinnerScalar : PyBinnerScalar {
    TBinnerScalar(int threads, std::string expr, double vmin, double vmax, uint64_t bins) : PyBinnerScalar(DT, FLIP, threads, std::move(expr), vmin, vmax, bins) {}
};

struct PyBinnerOrdinal : PyBinner {
    int64_t ordinal_count, min_value;
    bool allow_other, invert;
    PyBinnerOrdinal(int dt, bool fl, int threads_, std::string expr, int64_t ordinal_count, int64_t min_value, bool allow_other, bool invert)
        : ordinal_count(ordinal_count), min_value(min_value), allow_other(allow_other), invert(invert) {
        dtype = dt; flip = fl; threads = threads_; expression = std::move(expr);
        check(vxh_binner_ordinal_create(threads, dt, fl, ordinal_count, min_value, allow_other, invert, &h));
    }
};
template <int DT, bool FLIP>
struct TBinnerOrdinal : PyBinnerOrdinal {
    TBinnerOrdinal(int threads, std::string expr, int64_t ordinal_count, int64_t min_value, bool allow_other, bool invert)
        : PyBinnerOrdinal(DT, FLIP, threads, std::move(expr), ordinal_count, min_value, allow_other, invert) {}
};

// BinnerHash_<T>[_non_native](threads, expression, hashmap) — src/binner_hash.cpp:13-20, :148-171.  `hashmap` is
//   * a vaex_amd.hashset.ordered_set_<T> (what vaex.hash holds after install(); anything with `_binner_view()`): the reference's
//     cells [invalid, ordinal 0 .. ordinal len(set)-1, unused] with the null key / NaN among the ordinals (vxh_binner_hash_create_ref), or
//   * a bare device table (this module's ordered_set_<T>, used by vaex_amd.binned): cells [unknown, ordinal 0..N-1, null].
struct PyBinnerHash : PyBinner {
    py::object map_ref; // keeps the map alive
    PyBinnerHash(int dt, bool fl, int threads_, std::string expr, py::object map) : map_ref(std::move(map)) {
        dtype = dt; flip = fl; threads = threads_; expression = std::move(expr);
        if (py::isinstance<PyHashMap>(map_ref)) {
            if (fl) throw std::runtime_error("BinnerHash: byte-swapped keys need a vaex_amd.hashset set");
            PyHashMap &m = map_ref.cast<PyHashMap &>();
            check(vxh_binner_hash_create(threads, dt, m.h, &h));
            return;
        }
        if (!py::hasattr(map_ref, "_binner_view")) throw py::type_error("BinnerHash: expected a vaex_amd.hashset.ordered_set_<dtype> (or this module's ordered_set_<dtype>)");
        // (device table, public ordinal of every device ordinal or None, len(set), null_index, nan_index when the set saw a NaN else -1)
        py::tuple view = map_ref.attr("_binner_view")();
        PyHashMap &m = view[0].cast<PyHashMap &>();
        if (view[1].is_none()) {
            check(vxh_hashmap_set_public_ordinals(m.h, nullptr, 0));
        } else {
            auto perm = py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(view[1]);
            if (!perm) throw std::runtime_error("BinnerHash: the set's ordinals are not an int64 array");
            check(vxh_hashmap_set_public_ordinals(m.h, perm.data(), (uint64_t)perm.size()));
        }
        check(vxh_binner_hash_create_ref(threads, dt, fl ? 1 : 0, m.h, view[2].cast<uint64_t>(), view[3].cast<int64_t>(), view[4].cast<int64_t>(), &h));
    }
};
template <int DT, bool FLIP>
struct TBinnerHash : PyBinnerHash {
    TBinnerHash(int threads, std::string expr, py::object map) : PyBinnerHash(DT, FLIP, threads, std::move(expr), std::move(map)) {}
};

// ------------------------------------------------------------------------------------------
// grid
// ------------------------------------------------------------------------------------------
// page-locked bytes currently owned by result arrays (get_result): capped, see there
static std::atomic<size_t> g_pinned_result_bytes{0};
static constexpr size_t kPinnedResultBudget = (size_t)1 << 30;

struct PyAgg;
struct PyGrid {
    vxh_grid *h = nullptr;
    std::vector<PyBinner *> binners;
    explicit PyGrid(std::vector<PyBinner *> binners_) : binners(std::move(binners_)) {
        std::vector<vxh_binner *> hs;
        for (auto *b : binners) hs.push_back(b->h);
        check(vxh_grid_create(hs.data(), (int)hs.size(), &h));
    }
    ~PyGrid() { vxh_grid_destroy(h); }
    PyGrid(const PyGrid &) = delete;
    std::vector<uint64_t> shapes() const {
        std::vector<uint64_t> s(binners.size());
        vxh_grid_shapes(h, s.data());
        return s;
    }
    std::vector<uint64_t> strides() const {
        std::vector<uint64_t> s(binners.size());
        vxh_grid_strides(h, s.data());
        return s;
    }
    void bin(int thread, const std::vector<PyAgg *> &aggs, uint64_t length);
    void bin_all(int thread, const std::vector<PyAgg *> &aggs) {
        if (binners.empty()) throw std::runtime_error("no binners set and no length given"); // src/agg.hpp:78
        bin(thread, aggs, vxh_binner_data_length(binners[0]->h, thread));
    }
};

// ------------------------------------------------------------------------------------------
// aggregators
// ------------------------------------------------------------------------------------------
// device-side selection (include/vaex_hip.h "device-side selections"): built by vaex_amd.predicate from an expression string
struct PySelection {
    vxh_selection *h = nullptr;
    std::vector<int> dtypes;
    PySelection(int threads, const std::vector<int> &dtypes_, const std::vector<std::tuple<int, int, py::object>> &terms, uint32_t truth) : dtypes(dtypes_) {
        std::vector<vxh_sel_term> ts;
        for (auto &t : terms) {
            vxh_sel_term v{};
            v.column = std::get<0>(t);
            v.op = std::get<1>(t);
            const py::object &c = std::get<2>(t);
            if (py::isinstance<py::int_>(c) && !py::isinstance<py::bool_>(c)) {
                v.is_int = 1;
                v.ivalue = c.cast<int64_t>();
                v.value = (double)v.ivalue;
            } else {
                v.value = c.cast<double>();
                v.ivalue = (int64_t)v.value;
            }
            ts.push_back(v);
        }
        check(vxh_selection_create(threads, (int)dtypes.size(), dtypes.data(), (int)ts.size(), ts.data(), truth, &h));
    }
    // programs: {term index: [(op, column, value), ...]} — the terms whose left side is an arithmetic expression (vxh_selection_set_program)
    void set_programs(const std::map<int, std::vector<std::tuple<int, int, double>>> &programs) {
        for (auto &kv : programs) {
            std::vector<vxh_sel_step> steps;
            for (auto &st : kv.second) steps.push_back(vxh_sel_step{std::get<0>(st), std::get<1>(st), std::get<2>(st)});
            check(vxh_selection_set_program(h, kv.first, (int)steps.size(), steps.data()));
        }
    }
    HeldArrays held; // index = column
    ~PySelection() { vxh_selection_destroy(h); }
    PySelection(const PySelection &) = delete;
    void set_data(int thread, int column, const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        if (column < 0 || column >= (int)dtypes.size()) throw std::runtime_error("no such selection column");
        if (a.itemsize != kTypeSizes[dtypes[column]]) throw std::runtime_error("Itemsize of data and selection column are not equal");
        check(vxh_selection_set_data(h, thread, column, a.ptr, a.n, a.mem));
        held.hold(thread, column, ar, a);
    }
    // keep-mask of the slot's first n rows into `out` (a device uint8 array of >= n bytes, e.g. a torch tensor): vxh_selection_evaluate
    void evaluate(int thread, uint64_t n, const py::object &out) {
        ArrayRef o = resolve_array(out);
        if (o.mem != VXH_MEM_DEVICE || o.itemsize != 1 || o.n < ((n + 3) & ~(uint64_t)3)) throw std::runtime_error("Selection.evaluate: a device uint8 array of n bytes rounded up to 4");
        check(vxh_selection_evaluate(h, thread, n, (uint8_t *)o.ptr));
        held.hold(thread, 100, out, o); // (written by a kernel that may still be running when this returns)
    }
};

struct PyAgg {
    vxh_agg *h = nullptr;
    PyGrid *grid;
    int kind, dtype;
    py::object selection_ref; // keeps the attached selection alive
    HeldArrays held;          // index 0: data, 1: mask
    PyAgg(int kind, int dtype, bool flip, PyGrid *grid, int grids, int threads, uint32_t moment) : grid(grid), kind(kind), dtype(dtype) {
        check(vxh_agg_create(kind, dtype, flip, grid->h, grids, threads, moment, &h));
    }
    virtual ~PyAgg() { vxh_agg_destroy(h); }
    PyAgg(const PyAgg &) = delete;

    void set_data(int thread, const py::object &ar, size_t /*index*/) {
        ArrayRef a = resolve_array(ar);
        if (a.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and aggregator are not equal");
        check(vxh_agg_set_data(h, thread, a.ptr, a.n, a.mem));
        held.hold(thread, 0, ar, a);
    }
    void set_data_mask(int thread, const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        check(vxh_agg_set_data_mask(h, thread, (const uint8_t *)a.ptr, a.n, a.mem));
        held.hold(thread, 1, ar, a);
    }
    void clear_data_mask(int thread) {
        check(vxh_agg_clear_data_mask(h, thread));
        held.release(thread, 1);
    }
    void set_selection(const py::object &sel) {
        if (sel.is_none()) {
            check(vxh_agg_set_selection(h, nullptr));
            selection_ref = py::none();
        } else {
            check(vxh_agg_set_selection(h, sel.cast<PySelection *>()->h));
            selection_ref = sel;
        }
    }
    size_t bytes_used() const { return vxh_agg_bytes_used(h); }

    void merge(const std::vector<PyAgg *> &others) {
        std::vector<vxh_agg *> hs;
        for (auto *o : others) hs.push_back(o->h);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_agg_merge(h, hs.data(), (int)hs.size());
        }
        check(rc);
    }

    // fresh ndarray of `shapes`, dim 0 fastest (= numpy.array(self)[0] of the reference, src/agg_count.cpp:38-40)
    py::array get_result() {
        const int gdt = vxh_agg_grid_dtype(h);
        const ssize_t isz = kTypeSizes[gdt];
        std::vector<uint64_t> shp = grid->shapes(), str = grid->strides();
        std::vector<ssize_t> shape(shp.begin(), shp.end()), strides;
        for (auto s : str) strides.push_back((ssize_t)s * isz);
        const std::string fmt = kNumpyFormats[gdt];
        py::dtype np_dtype{fmt};
        // Page-locked memory from the library's host block cache for anything beyond a few pages: a device-to-host copy into
        // pageable memory is staged by the runtime (18 MB of a 128^3 count grid: 1.17 ms against 0.33 ms; the 0.5 MB grids of the
        // bench pass: ~60 us against ~15 us each).  The array owns the block and gives it back to the cache when it dies.
        //  Bounded: results a caller keeps alive hold at most kPinnedResultBudget of page-locked memory between them — beyond
        // that, and whenever the pinned allocation fails, the result is an ordinary pageable array (slower copy, never an error).
        const size_t bytes = (size_t)vxh_grid_length1d(grid->h) * (size_t)isz;
        py::array out;
        void *p = nullptr;
        if (bytes >= (64u << 10) && g_pinned_result_bytes.load() + bytes <= kPinnedResultBudget && vxh_host_alloc(bytes, &p) == 0 && p) {
            g_pinned_result_bytes += bytes;
            struct Block { void *p; size_t bytes; };
            py::capsule owner(new Block{p, bytes}, [](void *q) {
                Block *b = (Block *)q;
                vxh_host_free(b->p); // (back to the library's host block cache — an object that is never destroyed, so this is safe during interpreter shutdown)
                g_pinned_result_bytes -= b->bytes;
                delete b;
            });
            out = py::array(np_dtype, shape, strides, p, owner);
        } else {
            out = py::array(np_dtype, shape, strides);
        }
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_agg_result(h, out.mutable_data());
        }
        check(rc);
        return out;
    }

    // buffer protocol: (grids, *shapes), src/agg_base.hpp:106-125
    py::buffer_info buffer() {
        void *ptr = nullptr;
        check(vxh_agg_host_view(h, &ptr));
        const int gdt = vxh_agg_grid_dtype(h);
        const ssize_t isz = kTypeSizes[gdt];
        std::vector<uint64_t> shp = grid->shapes(), str = grid->strides();
        const size_t nd = shp.size();
        std::vector<ssize_t> shapes(nd + 1), strides(nd + 1);
        shapes[0] = vxh_agg_grids(h);
        for (size_t i = 0; i < nd; i++) {
            shapes[i + 1] = (ssize_t)shp[i];
            strides[i + 1] = (ssize_t)str[i] * isz;
        }
        strides[0] = nd ? strides[1] * shapes[1] : isz;
        if (nd) strides[0] = (ssize_t)vxh_grid_length1d(grid->h) * isz;
        return py::buffer_info(ptr, isz, kNumpyFormats[gdt], (ssize_t)nd + 1, shapes, strides);
    }

    // device pointer of the folded grid as an object with __cuda_array_interface__ (for RCCL all-reduce)
    py::dict device_grid_interface() {
        void *dptr = nullptr;
        int ddt = 0;
        check(vxh_agg_device_grid(h, &dptr, &ddt));
        static const char *typestr[VXH_DTYPE_COUNT] = {"<f8", "<f4", "<i8", "<i4", "<i2", "|i1", "<u8", "<u4", "<u2", "|u1", "|b1"};
        py::dict d;
        d["shape"] = py::make_tuple((uint64_t)vxh_grid_length1d(grid->h));
        d["typestr"] = typestr[ddt];
        d["data"] = py::make_tuple((uintptr_t)dptr, false);
        d["version"] = 3;
        d["strides"] = py::none();
        return d;
    }
    void device_touch() { check(vxh_agg_device_touch(h)); }
    void reset() { check(vxh_agg_reset(h)); }
};

void PyGrid::bin(int thread, const std::vector<PyAgg *> &aggs, uint64_t length) {
    collect_retired(thread);
    std::vector<vxh_agg *> hs;
    for (auto *a : aggs) hs.push_back(a->h);
    int rc;
    {
        py::gil_scoped_release release; // src/agg.hpp:94-99
        rc = vxh_grid_bin(h, thread, hs.data(), (int)hs.size(), length);
    }
    check(rc);
}

// AggFirst_<T>_<T2>[_non_native](grid, grids, threads, invert) — src/agg_first.cpp:165-178.  ONE class here; the 11 x 11 x 2
// names of the reference are made on demand by the module's __getattr__ (vaex looks them up by name: vaex/agg.py:279-285)
struct PyAggFirst {
    vxh_first *h = nullptr;
    PyGrid *grid;
    int dtype, dtype_order;
    PyAggFirst(PyGrid *grid, int grids, int threads, bool invert, int dtype, int dtype_order, bool flip) : grid(grid), dtype(dtype), dtype_order(dtype_order) {
        check(vxh_first_create(dtype, dtype_order, flip, grid->h, grids, threads, invert, &h));
    }
    HeldArrays held; // index 0 / 1: value / order column, 2: mask
    ~PyAggFirst() { vxh_first_destroy(h); }
    PyAggFirst(const PyAggFirst &) = delete;
    void set_data(int thread, const py::object &ar, size_t index) {
        ArrayRef a = resolve_array(ar);
        if (a.itemsize != kTypeSizes[index == 1 ? dtype_order : dtype]) throw std::runtime_error("Itemsize of data and aggregator are not equal");
        check(vxh_first_set_data(h, thread, (int)index, a.ptr, a.n, a.mem));
        held.hold(thread, index == 1 ? 1 : 0, ar, a);
    }
    void set_data_mask(int thread, const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        check(vxh_first_set_data_mask(h, thread, (const uint8_t *)a.ptr, a.n, a.mem));
        held.hold(thread, 2, ar, a);
    }
    void clear_data_mask(int thread) {
        check(vxh_first_set_data_mask(h, thread, nullptr, 0, VXH_MEM_HOST));
        held.release(thread, 2);
    }
    size_t bytes_used() const { return vxh_first_bytes_used(h); }
    // (values, masked, order) as arrays of `shapes`, dim 0 fastest
    py::tuple raw_result() {
        std::vector<uint64_t> shp = grid->shapes(), str = grid->strides();
        auto make = [&](int dt) {
            std::vector<ssize_t> shape(shp.begin(), shp.end()), strides;
            for (auto v : str) strides.push_back((ssize_t)v * kTypeSizes[dt]);
            return py::array(py::dtype(std::string(kNumpyFormats[dt])), shape, strides);
        };
        py::array values = make(dtype), masked = make(VXH_BOOL), order = make(dtype_order);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_first_result(h, values.mutable_data(), (uint8_t *)masked.mutable_data(), order.mutable_data());
        }
        check(rc);
        return py::make_tuple(values, masked, order);
    }
    // numpy.ma array like the reference (src/agg_first.cpp:60-117)
    py::object get_result() {
        py::tuple r = raw_result();
        return py::module::import("numpy.ma").attr("array")(r[0], py::arg("mask") = r[1]);
    }
};

// AggNUnique_<T>[_non_native](grid, grids, threads, dropmissing, dropnan) — src/agg_nunique.cpp:226-235 — and
// AggList_<T>_<T2>[_non_native](grid, grids, threads, dropnan, dropnull) — src/agg_list.cpp:223-233: ONE class each here, the
// reference's names come from the module's __getattr__
struct PyCollect {
    vxh_collect *h = nullptr;
    PyGrid *grid;
    int mode, dtype;
    PyCollect(PyGrid *grid, int grids, int threads, bool a, bool b, int mode, int dtype, bool flip) : grid(grid), mode(mode), dtype(dtype) {
        check(vxh_collect_create(mode, dtype, flip, grid->h, grids, threads, a, b, &h));
    }
    HeldArrays held; // index 0: data, 1: mask, 2: selection mask
    ~PyCollect() { vxh_collect_destroy(h); }
    PyCollect(const PyCollect &) = delete;
    void set_data(int thread, const py::object &ar, size_t index) {
        if (index != 0) return; // (AggList's second column is registered and never read: src/agg_list.cpp:91)
        ArrayRef a = resolve_array(ar);
        if (a.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and aggregator are not equal");
        check(vxh_collect_set_data(h, thread, a.ptr, a.n, a.mem));
        held.hold(thread, 0, ar, a);
    }
    void set_data_mask(int thread, const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        check(vxh_collect_set_data_mask(h, thread, (const uint8_t *)a.ptr, a.n, a.mem));
        held.hold(thread, 1, ar, a);
    }
    void clear_data_mask(int thread) {
        check(vxh_collect_set_data_mask(h, thread, nullptr, 0, VXH_MEM_HOST));
        held.release(thread, 1);
    }
    void set_selection_mask(int thread, const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        check(vxh_collect_set_selection_mask(h, thread, (const uint8_t *)a.ptr, a.n, a.mem));
        held.hold(thread, 2, ar, a);
    }
    void clear_selection_mask(int thread) {
        check(vxh_collect_set_selection_mask(h, thread, nullptr, 0, VXH_MEM_HOST));
        held.release(thread, 2);
    }
    void bin(int thread, uint64_t length) {
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_collect_bin(h, thread, length);
        }
        check(rc);
    }
    // nunique: int64 array of the grid's shape, transposed like the reference's (src/agg_nunique.cpp:40-44)
    py::object nunique_result() {
        std::vector<uint64_t> shp = grid->shapes(), str = grid->strides();
        std::vector<ssize_t> shape(shp.begin(), shp.end()), strides;
        for (auto v : str) strides.push_back((ssize_t)v * 8);
        py::array out(py::dtype("int64"), shape, strides);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_collect_nunique_result(h, (int64_t *)out.mutable_data());
        }
        check(rc);
        return out;
    }
    // list: (offsets int64[cells + 1], values[flat]) — what the reference hands to vaex.arrow.convert.list_from_arrays
    py::tuple list_arrays() {
        uint64_t cells = 1;
        for (auto v : grid->shapes()) cells *= v;
        py::array_t<int64_t> offsets((ssize_t)cells + 1);
        uint64_t flat = 0;
        check(vxh_collect_list_result(h, offsets.mutable_data(), nullptr, &flat));
        py::array values(py::dtype(std::string(kNumpyFormats[dtype])), std::vector<ssize_t>{(ssize_t)flat});
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_collect_list_result(h, offsets.mutable_data(), values.mutable_data(), &flat);
        }
        check(rc);
        return py::make_tuple(offsets, values);
    }
    // (values as uint64 canonical bits, cells uint32, missing rows per cell int64, NaN rows per cell int64): vxh_collect_pairs
    py::tuple pairs() {
        uint64_t cells = 1;
        for (auto v : grid->shapes()) cells *= v;
        uint64_t n = 0;
        py::array_t<int64_t> nulls((ssize_t)cells), nans((ssize_t)cells);
        check(vxh_collect_pairs(h, &n, nullptr, nullptr, nulls.mutable_data(), nans.mutable_data()));
        py::array_t<uint64_t> values((ssize_t)n);
        py::array_t<uint32_t> cell((ssize_t)n);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_collect_pairs(h, &n, values.mutable_data(), cell.mutable_data(), nulls.mutable_data(), nans.mutable_data());
        }
        check(rc);
        if ((uint64_t)values.size() != n) throw std::runtime_error("collector changed while its pairs were read");
        return py::make_tuple(values, cell, nulls, nans);
    }
    void merge_pairs(py::array_t<uint64_t, py::array::c_style | py::array::forcecast> values, py::array_t<uint32_t, py::array::c_style | py::array::forcecast> cell,
                     py::array_t<int64_t, py::array::c_style | py::array::forcecast> nulls, py::array_t<int64_t, py::array::c_style | py::array::forcecast> nans) {
        uint64_t cells = 1;
        for (auto v : grid->shapes()) cells *= v;
        if (values.size() != cell.size() || (uint64_t)nulls.size() != cells || (uint64_t)nans.size() != cells) throw std::runtime_error("merge_pairs: inconsistent arguments");
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_collect_merge_pairs(h, (uint64_t)values.size(), values.data(), cell.data(), nulls.data(), nans.data());
        }
        check(rc);
    }
    py::object get_result() {
        if (mode == 0) return nunique_result();
        py::tuple r = list_arrays();
        return py::module::import("vaex.arrow.convert").attr("list_from_arrays")(r[0], r[1]); // src/agg_list.cpp:81-83
    }
};

template <int KIND, int DT, bool FLIP>
struct TAgg : PyAgg {
    TAgg(PyGrid *grid, int grids, int threads) : PyAgg(KIND, DT, FLIP, grid, grids, threads, 0) {}
    TAgg(PyGrid *grid, int grids, int threads, uint32_t moment) : PyAgg(KIND, DT, FLIP, grid, grids, threads, moment) {}
};

// ------------------------------------------------------------------------------------------
// registration
// ------------------------------------------------------------------------------------------
template <int DT, bool FLIP>
void add_binners(py::module &m, py::class_<PyBinnerScalar, PyBinner> &scalar_base, py::class_<PyBinnerOrdinal, PyBinner> &ordinal_base) {
    const std::string postfix = std::string(kTypeNames[DT]) + (FLIP ? "_non_native" : "");
    {
        using T = TBinnerScalar<DT, FLIP>;
        py::class_<T, PyBinnerScalar>(m, ("BinnerScalar_" + postfix).c_str())
            .def(py::init<int, std::string, double, double, uint64_t>())
            .def("copy", [](const T &b) { return new T(b); })
            .def(py::pickle([](const T &b) { return py::make_tuple(b.threads, b.expression, b.vmin, b.vmax, b.bins); },
                            [](py::tuple t) {
                                if (t.size() != 5) throw std::runtime_error("Invalid state!");
                                return new T(t[0].cast<int>(), t[1].cast<std::string>(), t[2].cast<double>(), t[3].cast<double>(), t[4].cast<uint64_t>());
                            }));
    }
    {
        using T = TBinnerOrdinal<DT, FLIP>;
        py::class_<T, PyBinnerOrdinal>(m, ("BinnerOrdinal_" + postfix).c_str())
            .def(py::init<int, std::string, int64_t, int64_t, bool, bool>(), py::arg("threads"), py::arg("expression"), py::arg("ordinal_count"), py::arg("min_value") = 0, py::arg("allow_other") = false,
                 py::arg("invert") = false)
            .def("copy", [](const T &b) { return new T(b); })
            .def(py::pickle([](const T &b) { return py::make_tuple(b.threads, b.expression, b.ordinal_count, b.min_value, b.allow_other, b.invert); },
                            [](py::tuple t) {
                                if (t.size() != 6) throw std::runtime_error("Invalid state!");
                                return new T(t[0].cast<int>(), t[1].cast<std::string>(), t[2].cast<int64_t>(), t[3].cast<int64_t>(), t[4].cast<bool>(), t[5].cast<bool>());
                            }));
    }
}

template <int DT, bool FLIP>
void add_aggs(py::module &m, py::class_<PyAgg> &base) {
    const std::string postfix = std::string(kTypeNames[DT]) + (FLIP ? "_non_native" : "");
#define VXH_AGG3(KIND, NAME)                                                                                           \
    py::class_<TAgg<KIND, DT, FLIP>, PyAgg>(m, (std::string(NAME) + postfix).c_str(), py::buffer_protocol())           \
        .def(py::init<PyGrid *, int, int>(), py::keep_alive<1, 2>())                                                   \
        .def_buffer([](TAgg<KIND, DT, FLIP> &a) { return a.buffer(); });
    VXH_AGG3(VXH_AGG_COUNT, "AggCount_")
    VXH_AGG3(VXH_AGG_SUM, "AggSum_")
    VXH_AGG3(VXH_AGG_MIN, "AggMin_")
    VXH_AGG3(VXH_AGG_MAX, "AggMax_")
#undef VXH_AGG3
    py::class_<TAgg<VXH_AGG_SUM_MOMENT, DT, FLIP>, PyAgg>(m, ("AggSumMoment_" + postfix).c_str(), py::buffer_protocol())
        .def(py::init<PyGrid *, int, int, uint32_t>(), py::keep_alive<1, 2>())
        .def_buffer([](TAgg<VXH_AGG_SUM_MOMENT, DT, FLIP> &a) { return a.buffer(); });
}

template <int DT>
void add_hash(py::module &m, py::class_<PyHashMap> &map_base) {
    py::class_<THashMap<DT>, PyHashMap>(m, (std::string("ordered_set_") + kTypeNames[DT]).c_str()).def(py::init<uint64_t>(), py::arg("capacity_hint") = 0);
}

template <int DT, bool FLIP>
void add_binner_hash(py::module &m, py::class_<PyBinnerHash, PyBinner> &binner_base) {
    typedef TBinnerHash<DT, FLIP> Type;
    py::class_<Type, PyBinnerHash>(m, (std::string("BinnerHash_") + kTypeNames[DT] + (FLIP ? "_non_native" : "")).c_str())
        .def(py::init<int, std::string, py::object>())
        .def("copy", [](const Type &b) { return new Type(b); })
        .def(py::pickle([](const Type &b) { return py::make_tuple(b.threads, b.expression, b.map_ref); }, // src/binner_hash.cpp:159-170
                        [](py::tuple t) {
                            if (t.size() != 3) throw std::runtime_error("Invalid state!");
                            return new Type(t[0].cast<int>(), t[1].cast<std::string>(), t[2]);
                        }));
}

template <int DT>
void add_type(py::module &m, py::class_<PyBinnerScalar, PyBinner> &sb, py::class_<PyBinnerOrdinal, PyBinner> &ob, py::class_<PyAgg> &ab) {
    add_binners<DT, false>(m, sb, ob);
    add_binners<DT, true>(m, sb, ob);
    add_aggs<DT, false>(m, ab);
    add_aggs<DT, true>(m, ab);
}

} // namespace

PYBIND11_MODULE(superagg, m) {
    m.doc() = "MI355X (HIP) implementation of the vaex.superagg class surface: binned statistics / groupby aggregation";
    m.attr("__hip__") = true;
    m.def("abi_version", &vxh_abi_version);
    m.attr("AUX_SLOT") = (int)VXH_AUX_SLOT; // first thread slot outside the host pool's indices (include/vaex_hip.h)
    m.def("device_count", []() { int n = 0; check(vxh_device_count(&n)); return n; });
    m.def("set_device", [](int d) { check(vxh_set_device(d)); });
    m.def("warmup", []() { py::gil_scoped_release r; check(vxh_warmup()); }, "load the kernels' code objects and create thread slot 0 now (vaex_amd.install() does)");
    m.def("synchronize", []() {
        { py::gil_scoped_release r; check(vxh_synchronize()); }
        collect_all_retired();
    });
    // lifetime of device columns (include/vaex_hip.h "Data pointers"): the slot's streams may still read them after Grid.bin returned
    m.def("slot_busy", [](int thread) { int busy = 0; check(vxh_slot_busy(thread, &busy)); return busy != 0; }, py::arg("thread") = 0);
    m.def("slot_wait", [](int thread) {
        { py::gil_scoped_release r; check(vxh_slot_wait(thread)); }
        collect_retired(thread);
    }, py::arg("thread") = 0);
    m.def("wait_stream", [](int thread, uintptr_t stream) { check(vxh_slot_wait_stream(thread, (void *)stream)); }, py::arg("thread"), py::arg("stream"),
          "order slot `thread`'s next work after everything enqueued so far on the hipStream_t `stream` (e.g. torch.cuda.current_stream().cuda_stream)");
    m.def("retired_device_arrays", [](int thread) { auto &all = retired_refs(); auto it = all.find(thread); return it == all.end() ? (size_t)0 : it->second.size(); }, py::arg("thread") = 0,
          "device arrays the shim still holds for slot `thread` although their holders have let go of them (their kernels may still be running)");
    m.def("config_set", [](const std::string &k, int64_t v) { check(vxh_config_set(k.c_str(), v)); });
    m.def("config_get", [](const std::string &k) { int64_t v = 0; check(vxh_config_get(k.c_str(), &v)); return v; });
    // device column cache (include/vaex_hip.h "chunk feeder and device column cache"): the array's memory is declared immutable
    // until cache_unregister(array); returns whether it got page-locked
    // upload(host_array, device_array, threads=0): the host array's bytes into the device array (same byte length), several copy threads
    m.def("upload", [](const py::object &src, const py::object &dst, int threads) {
        ArrayRef a = resolve_array(src), d = resolve_array(dst);
        if (a.mem != VXH_MEM_HOST || d.mem != VXH_MEM_DEVICE) throw std::runtime_error("upload: (host array, device array)");
        if ((uint64_t)a.n * a.itemsize != (uint64_t)d.n * d.itemsize) throw std::runtime_error("upload: the arrays differ in byte length");
        int rc;
        { py::gil_scoped_release r; rc = vxh_upload(a.ptr, (void *)d.ptr, (uint64_t)a.n * a.itemsize, threads); }
        check(rc);
    }, py::arg("src"), py::arg("dst"), py::arg("threads") = 0);
    m.def("cache_register", [](const py::object &ar, bool pin) {
        ArrayRef a = resolve_array(ar);
        if (a.mem != VXH_MEM_HOST) throw std::runtime_error("cache_register: host arrays only");
        int pinned = 0, rc;
        { py::gil_scoped_release r; rc = vxh_cache_register(a.ptr, (uint64_t)a.n * a.itemsize, pin ? VXH_CACHE_PIN : 0, &pinned); }
        check(rc);
        return pinned != 0;
    }, py::arg("array"), py::arg("pin") = true);
    m.def("cache_unregister", [](const py::object &ar) {
        ArrayRef a = resolve_array(ar);
        int rc;
        { py::gil_scoped_release r; rc = vxh_cache_unregister(a.ptr); }
        check(rc);
    });
    m.def("cache_clear", []() { py::gil_scoped_release r; check(vxh_cache_clear()); });
    m.def("cache_stats", []() {
        uint64_t v[6];
        check(vxh_cache_stats(v));
        py::dict d;
        d["bytes"] = v[0]; d["chunks"] = v[1]; d["hits"] = v[2]; d["misses"] = v[3]; d["evictions"] = v[4]; d["ranges"] = v[5];
        return d;
    });
    m.def("last_kernel", [](int thread) { return std::string(vxh_last_kernel(thread)); }, py::arg("thread") = 0);
    m.def("slot_set_stream", [](int thread, uintptr_t stream) { check(vxh_slot_set_stream(thread, (void *)stream)); });
    m.def("timer_start", [](int thread) { check(vxh_timer_start(thread)); }, py::arg("thread") = 0);
    m.def("timer_stop", [](int thread) { float ms = 0; { py::gil_scoped_release r; check(vxh_timer_stop(thread, &ms)); } return ms; }, py::arg("thread") = 0);
    m.def("timer_kernels_ms", [](int thread) { float ms = 0; { py::gil_scoped_release r; check(vxh_timer_kernels_ms(thread, &ms)); } return ms; }, py::arg("thread") = 0);
    m.def("minmax", [](const py::object &ar, const py::object &mask, int dtype, bool flip) {
        ArrayRef a = resolve_array(ar);
        if (a.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and dtype are not equal");
        const uint8_t *mp = nullptr;
        if (!mask.is_none()) { ArrayRef mk = resolve_array(mask); check_mask(a, mk); mp = (const uint8_t *)mk.ptr; }
        double out[2];
        int rc;
        { py::gil_scoped_release r; rc = vxh_minmax(dtype, flip, a.ptr, mp, a.n, a.mem, out); }
        check(rc);
        return py::make_tuple(out[0], out[1]);
    }, py::arg("data"), py::arg("mask") = py::none(), py::arg("dtype") = 0, py::arg("flip") = false);

    m.def("minmax_int", [](const py::object &ar, const py::object &mask, int dtype, bool flip) {
        ArrayRef a = resolve_array(ar);
        if (a.itemsize != kTypeSizes[dtype]) throw std::runtime_error("Itemsize of data and dtype are not equal");
        const uint8_t *mp = nullptr;
        if (!mask.is_none()) { ArrayRef mk = resolve_array(mask); check_mask(a, mk); mp = (const uint8_t *)mk.ptr; }
        int64_t out[2];
        int rc;
        { py::gil_scoped_release r; rc = vxh_minmax_int(dtype, flip, a.ptr, mp, a.n, a.mem, out); }
        check(rc);
        return py::make_tuple(out[0], out[1]);
    }, py::arg("data"), py::arg("mask") = py::none(), py::arg("dtype") = 2, py::arg("flip") = false);

    // one pass over an int64 key column and a float64 value column on the device: (key min, key max, NaN values)
    m.def("scan_key_value", [](const py::object &keys, const py::object &values) {
        ArrayRef k = resolve_array(keys), v = resolve_array(values);
        if (k.mem != VXH_MEM_DEVICE || v.mem != VXH_MEM_DEVICE || k.itemsize != 8 || v.itemsize != 8 || k.n != v.n) throw std::runtime_error("scan_key_value: two device columns of 8-byte elements and equal length");
        int64_t out[3];
        int rc;
        { py::gil_scoped_release r; rc = vxh_scan_key_value((const int64_t *)k.ptr, (const double *)v.ptr, k.n, out); }
        check(rc);
        return py::make_tuple(out[0], out[1], out[2]);
    }, py::arg("keys"), py::arg("values"));

    // finishers on the device: finish([(op, agg0, agg1 | None, agg2 | None), ...], present=None, first=0, n=None)
    // -> (list of 1-d ndarrays in pinned host memory, index ndarray | None)
    m.def("finish", [](const std::vector<py::tuple> &specs, const py::object &present, uint64_t first, const py::object &n_obj, bool want_index) {
        const int n_out = (int)specs.size();
        if (n_out < 1) throw std::runtime_error("finish: no result columns");
        std::vector<int> ops(n_out);
        std::vector<vxh_agg *> a0(n_out, nullptr), a1(n_out, nullptr), a2(n_out, nullptr);
        std::vector<std::string> fmt(n_out);
        for (int j = 0; j < n_out; j++) {
            const py::tuple &t = specs[j];
            if (t.size() < 2) throw std::runtime_error("finish: (op, agg0[, agg1[, agg2]]) expected");
            ops[j] = t[0].cast<int>();
            a0[j] = t[1].cast<PyAgg &>().h;
            if (t.size() > 2 && !t[2].is_none()) a1[j] = t[2].cast<PyAgg &>().h;
            if (t.size() > 3 && !t[3].is_none()) a2[j] = t[3].cast<PyAgg &>().h;
            fmt[j] = "d";
            if (ops[j] == VXH_FIN_COPY) {
                const int gdt = vxh_agg_grid_dtype(a0[j]);
                fmt[j] = (gdt == VXH_F64 || gdt == VXH_F32) ? "d" : ((gdt == VXH_I64 || gdt == VXH_I32 || gdt == VXH_I16 || gdt == VXH_I8) ? "q" : "Q");
            }
        }
        PyAgg &first_agg = specs[0][1].cast<PyAgg &>();
        const uint64_t cells = vxh_grid_length1d(first_agg.grid->h);
        const uint64_t n = n_obj.is_none() ? cells - first : n_obj.cast<uint64_t>();
        vxh_agg *pres = present.is_none() ? nullptr : present.cast<PyAgg &>().h;
        // result columns live in pinned host memory owned by the arrays (freed with them)
        auto pinned = [](uint64_t elems, const std::string &f) { return pinned_array(elems, f); };
        std::vector<py::array> cols;
        std::vector<void *> outs(n_out);
        for (int j = 0; j < n_out; j++) {
            cols.push_back(pinned(n, fmt[j]));
            outs[j] = cols.back().mutable_data();
        }
        py::object index = py::none();
        int64_t *index_ptr = nullptr;
        if (want_index) {
            py::array ia = pinned(n, "q");
            index_ptr = (int64_t *)ia.mutable_data();
            index = ia;
        }
        uint64_t kept = 0;
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_finish(n_out, ops.data(), a0.data(), a1.data(), a2.data(), pres, first, n, outs.data(), index_ptr, &kept);
        }
        check(rc);
        py::list out;
        py::slice sl(0, (ssize_t)kept, 1);
        for (auto &c : cols) out.append(c[sl]);
        if (want_index) index = index[sl];
        return py::make_tuple(out, index);
    }, py::arg("specs"), py::arg("present") = py::none(), py::arg("first") = 0, py::arg("n") = py::none(), py::arg("want_index") = true);
    m.attr("FIN_COPY") = (int)VXH_FIN_COPY;
    m.attr("FIN_MEAN") = (int)VXH_FIN_MEAN;
    m.attr("FIN_VAR") = (int)VXH_FIN_VAR;
    m.attr("FIN_STD") = (int)VXH_FIN_STD;

    py::class_<PyGroupBy>(m, "GroupByResult")
        .def("__len__", &PyGroupBy::size)
        .def("column", &PyGroupBy::column, py::arg("which"), py::arg("value_index") = 0)
        .def("info", &PyGroupBy::info);
    m.attr("GB_KEYS") = (int)VXH_GB_KEYS;
    m.attr("GB_ROWS") = (int)VXH_GB_ROWS;
    m.attr("GB_COUNT") = (int)VXH_GB_COUNT;
    m.attr("GB_SUM") = (int)VXH_GB_SUM;
    m.attr("GB_SUM2") = (int)VXH_GB_SUM2;
    m.attr("GB_MEAN") = (int)VXH_GB_MEAN;
    m.attr("GB_VAR") = (int)VXH_GB_VAR;
    m.attr("GB_STD") = (int)VXH_GB_STD;
    // groupby_run(keys, [v0, v1], key_dtype, groups_hint=0, max_groups=0): keys any integer array (host or device), values
    // float64 arrays living where the keys live
    m.def("sample_heavy_keys", [](const py::object &keys, int key_dtype, uint32_t sample, uint32_t min_count, int max_keys) {
        ArrayRef k = resolve_array(keys);
        if (k.mem != VXH_MEM_DEVICE) throw std::runtime_error("sample_heavy_keys: a device-resident key column");
        if (k.itemsize != kTypeSizes[key_dtype]) throw std::runtime_error("Itemsize of keys and key dtype are not equal");
        py::array_t<int64_t> out((size_t)std::max(max_keys, 0));
        int n_out = 0, rc;
        { py::gil_scoped_release r; rc = vxh_sample_heavy_keys(key_dtype, k.ptr, k.n, sample, min_count, max_keys, out.mutable_data(), &n_out); }
        check(rc);
        out.resize({(size_t)n_out});
        return out;
    }, py::arg("keys"), py::arg("key_dtype") = (int)VXH_I64, py::arg("sample") = 1u << 17, py::arg("min_count") = 8, py::arg("max_keys") = 128,
       "heavy hitters of a device-resident integer key column from a strided sample (vxh_sample_heavy_keys)");
    m.def("groupby_run", [](const py::object &keys, const std::vector<py::object> &values, int key_dtype, uint64_t hint, uint64_t max_groups, const py::object &keep, const py::object &key_range, const py::object &heavy, const py::object &pred) {
        // pred = ([(value index, vxh_cmp op, constant), ...], truth): the filter as terms over the call's own value columns (vxh_groupby_run_selected)
        std::vector<vxh_groupby_term> terms;
        uint32_t truth = 0;
        if (!pred.is_none()) {
            auto pr = pred.cast<std::pair<std::vector<std::tuple<int, int, double>>, uint32_t>>();
            for (auto &t : pr.first) terms.push_back(vxh_groupby_term{std::get<0>(t), std::get<1>(t), std::get<2>(t)});
            truth = pr.second;
        }
        std::vector<int64_t> hv; // heavy keys peeled inside the pass (vxh_groupby_run_peeled)
        if (!heavy.is_none()) for (auto item : py::array_t<int64_t, py::array::c_style | py::array::forcecast>::ensure(heavy).cast<std::vector<int64_t>>()) hv.push_back(item);
        int64_t key_min = 1, key_max = 0; // (unknown)
        if (!key_range.is_none()) {
            auto kr = key_range.cast<std::pair<int64_t, int64_t>>();
            key_min = kr.first; key_max = kr.second;
        }
        ArrayRef k = resolve_array(keys);
        const uint8_t *keep_ptr = nullptr;
        if (!keep.is_none()) { // uint8 keep-mask where the keys live (vxh_groupby_run_kept)
            ArrayRef km = resolve_array(keep);
            if (km.itemsize != 1 || km.mem != k.mem || km.n < k.n) throw std::runtime_error("groupby: the keep-mask must be one byte per row and live where the keys live");
            keep_ptr = (const uint8_t *)km.ptr;
        }
        if (key_dtype < 0 || key_dtype >= VXH_DTYPE_COUNT || k.itemsize != kTypeSizes[key_dtype]) throw std::runtime_error("Itemsize of the keys and key dtype are not equal");
        std::vector<const void *> vp;
        for (const auto &v : values) {
            ArrayRef a = resolve_array(v);
            if (a.itemsize != 8) throw std::runtime_error("groupby: float64 value columns only");
            if (a.mem != k.mem || a.n < k.n) throw std::runtime_error("groupby: value columns must live where the keys live and be as long");
            vp.push_back(a.ptr);
        }
        auto res = std::make_unique<PyGroupBy>();
        res->nv = (int)vp.size();
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_groupby_run_selected(key_dtype, k.ptr, (int)vp.size(), vp.data(), keep_ptr, k.n, k.mem, hint, max_groups, key_min, key_max, hv.data(), (int)hv.size(),
                                          (int)terms.size(), terms.data(), truth, &res->h);
        }
        check(rc);
        return res;
    }, py::arg("keys"), py::arg("values"), py::arg("key_dtype") = (int)VXH_I64, py::arg("groups_hint") = 0, py::arg("max_groups") = 0, py::arg("keep") = py::none(), py::arg("key_range") = py::none(), py::arg("heavy") = py::none(),
       py::arg("pred") = py::none());
    m.attr("GROUPBY_PRED") = 1; // groupby_run takes `pred`
    // groupby_merge(keys, rows, [count_j], [sum_j], [sum2_j]): partial results (host arrays) -> one result
    m.def("groupby_merge", [](py::array_t<int64_t, py::array::c_style | py::array::forcecast> keys, py::array_t<int64_t, py::array::c_style | py::array::forcecast> rows,
                              const std::vector<py::array_t<int64_t, py::array::c_style | py::array::forcecast>> &counts,
                              const std::vector<py::array_t<double, py::array::c_style | py::array::forcecast>> &sums,
                              const std::vector<py::array_t<double, py::array::c_style | py::array::forcecast>> &sums2, uint64_t hint) {
        const uint64_t n = (uint64_t)keys.size();
        const size_t nv = counts.size();
        if (nv < 1 || sums.size() != nv || sums2.size() != nv || (uint64_t)rows.size() != n) throw std::runtime_error("groupby_merge: inconsistent arguments");
        std::vector<const int64_t *> cp;
        std::vector<const double *> sp, s2p;
        for (size_t v = 0; v < nv; v++) {
            if ((uint64_t)counts[v].size() != n || (uint64_t)sums[v].size() != n || (uint64_t)sums2[v].size() != n) throw std::runtime_error("groupby_merge: columns differ in length");
            cp.push_back(counts[v].data()); sp.push_back(sums[v].data()); s2p.push_back(sums2[v].data());
        }
        auto res = std::make_unique<PyGroupBy>();
        res->nv = (int)nv;
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_groupby_merge((int)nv, keys.data(), rows.data(), cp.data(), sp.data(), s2p.data(), n, hint, &res->h);
        }
        check(rc);
        return res;
    }, py::arg("keys"), py::arg("rows"), py::arg("counts"), py::arg("sums"), py::arg("sums2"), py::arg("groups_hint") = 0);

    py::class_<PyDeviceArray>(m, "DeviceArray")
        .def("__len__", [](const PyDeviceArray &a) { return a.n; })
        .def_property_readonly("__cuda_array_interface__", &PyDeviceArray::interface)
        .def_property_readonly("dtype", [](const PyDeviceArray &a) { return py::dtype(kNumpyFormats[a.dtype]); });
    // packed multi-key group key (vxh_pack_keys): integer columns, host or device -> int64 device array
    m.def("pack_keys", [](const std::vector<py::object> &columns, const std::vector<int> &dtypes, const std::vector<int64_t> &mins, const std::vector<int64_t> &mults) {
        const size_t nk = columns.size();
        if (!nk || dtypes.size() != nk || mins.size() != nk || mults.size() != nk) throw std::runtime_error("pack_keys: one dtype, minimum and multiplier per column");
        std::vector<const void *> ptrs;
        std::vector<int> mems;
        uint64_t n = 0;
        for (size_t k = 0; k < nk; k++) {
            ArrayRef a = resolve_array(columns[k]);
            if (a.itemsize != kTypeSizes[dtypes[k]]) throw std::runtime_error("pack_keys: itemsize of a column and its dtype differ");
            if (k && a.n != n) throw std::runtime_error("pack_keys: columns differ in length");
            n = a.n;
            ptrs.push_back(a.ptr);
            mems.push_back(a.mem);
        }
        auto out = std::make_unique<PyDeviceArray>(n, VXH_I64);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_pack_keys((int)nk, ptrs.data(), dtypes.data(), mems.data(), mins.data(), mults.data(), n, (int64_t *)out->p);
        }
        check(rc);
        return out;
    });
    // codes of a key with missing values / a float key (mode 0 -> int64), or a value column with NaN where missing (mode 1 -> float64): vxh_code_column
    m.def("code_column", [](const py::object &data, const py::object &mask, int dtype, int mode, int64_t null_code, int64_t nan_code, bool flip) {
        ArrayRef d = resolve_array(data);
        if (d.itemsize != kTypeSizes[dtype]) throw std::runtime_error("code_column: itemsize of the column and its dtype differ");
        ArrayRef k{};
        const bool has_mask = !mask.is_none();
        if (has_mask) {
            k = resolve_array(mask);
            if (k.itemsize != 1 || k.n != d.n) throw std::runtime_error("code_column: the mask is one byte per row of the column");
        }
        auto out = std::make_unique<PyDeviceArray>(d.n, mode == VXH_CODE_KEY ? VXH_I64 : VXH_F64);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_code_column(mode, dtype, d.ptr, d.mem, has_mask ? (const uint8_t *)k.ptr : nullptr, has_mask ? k.mem : VXH_MEM_HOST, flip ? 1 : 0, d.n, null_code, nan_code, out->p);
        }
        check(rc);
        return out;
    }, py::arg("data"), py::arg("mask"), py::arg("dtype"), py::arg("mode"), py::arg("null_code") = 0, py::arg("nan_code") = 0, py::arg("flip") = false);
    // row-wise product of two float64 columns (vxh_product_f64) -> float64 device array
    m.def("product", [](const py::object &a, const py::object &b) {
        ArrayRef x = resolve_array(a), y = resolve_array(b);
        if (x.itemsize != 8 || y.itemsize != 8) throw std::runtime_error("product: float64 columns");
        if (x.n != y.n) throw std::runtime_error("product: columns differ in length");
        auto out = std::make_unique<PyDeviceArray>(x.n, VXH_F64);
        int rc;
        {
            py::gil_scoped_release release;
            rc = vxh_product_f64((const double *)x.ptr, x.mem, (const double *)y.ptr, y.mem, x.n, (double *)out->p);
        }
        check(rc);
        return out;
    });
    py::class_<PySelection>(m, "Selection")
        .def(py::init<int, const std::vector<int> &, const std::vector<std::tuple<int, int, py::object>> &, uint32_t>(), py::arg("threads"), py::arg("dtypes"), py::arg("terms"), py::arg("truth"))
        .def("set_programs", &PySelection::set_programs, py::arg("programs"))
        .def("set_data", &PySelection::set_data)
        .def("evaluate", &PySelection::evaluate, py::arg("thread"), py::arg("n"), py::arg("out"));
    m.attr("SEL_COL") = (int)VXH_SEL_COL; m.attr("SEL_CONST") = (int)VXH_SEL_CONST; m.attr("SEL_ADD") = (int)VXH_SEL_ADD; m.attr("SEL_SUB") = (int)VXH_SEL_SUB;
    m.attr("SEL_MUL") = (int)VXH_SEL_MUL; m.attr("SEL_DIV") = (int)VXH_SEL_DIV; m.attr("SEL_NEG") = (int)VXH_SEL_NEG; m.attr("SEL_SQUARE") = (int)VXH_SEL_SQUARE;
    m.attr("SEL_SQRT") = (int)VXH_SEL_SQRT; m.attr("SEL_ABS") = (int)VXH_SEL_ABS;
    m.attr("SEL_LT") = (int)VXH_SEL_LT; m.attr("SEL_LE") = (int)VXH_SEL_LE; m.attr("SEL_GT") = (int)VXH_SEL_GT; m.attr("SEL_GE") = (int)VXH_SEL_GE; m.attr("SEL_EQ") = (int)VXH_SEL_EQ; m.attr("SEL_NE") = (int)VXH_SEL_NE;
    m.attr("CMP_LT") = (int)VXH_CMP_LT; m.attr("CMP_LE") = (int)VXH_CMP_LE; m.attr("CMP_GT") = (int)VXH_CMP_GT;
    m.attr("CMP_GE") = (int)VXH_CMP_GE; m.attr("CMP_EQ") = (int)VXH_CMP_EQ; m.attr("CMP_NE") = (int)VXH_CMP_NE;
    py::class_<PyAgg> aggregator(m, "Aggregator", py::buffer_protocol());
    // multi-GPU reduce on RCCL directly (vxh_comm_*, vxh_allreduce): Comm(n_ranks, rank, id) with rank 0's comm_unique_id()
    struct PyComm {
        vxh_comm *h = nullptr;
        ~PyComm() { vxh_comm_destroy(h); }
    };
    m.def("comm_unique_id", []() {
        char id[VXH_COMM_ID_BYTES];
        check(vxh_comm_unique_id(id));
        return py::bytes(id, VXH_COMM_ID_BYTES);
    });
    py::class_<PyComm>(m, "Comm")
        .def(py::init([](int n_ranks, int rank, const py::bytes &id) {
            const std::string raw = id;
            if (raw.size() != VXH_COMM_ID_BYTES) throw std::runtime_error("Comm: the id is comm_unique_id()'s 128 bytes");
            auto c = std::make_unique<PyComm>();
            int rc;
            {
                py::gil_scoped_release release; // (blocks until every rank has called it)
                rc = vxh_comm_init(n_ranks, rank, raw.data(), &c->h);
            }
            check(rc);
            return c;
        }), py::arg("n_ranks"), py::arg("rank"), py::arg("id"))
        .def_property_readonly("size", [](const PyComm &c) { return vxh_comm_size(c.h); })
        .def_property_readonly("rank", [](const PyComm &c) { return vxh_comm_rank(c.h); })
        .def("allreduce", [](PyComm &c, const std::vector<PyAgg *> &aggs) {
            std::vector<vxh_agg *> hs;
            for (auto *a : aggs) hs.push_back(a->h);
            int rc;
            {
                py::gil_scoped_release release;
                rc = vxh_allreduce(hs.data(), (int)hs.size(), c.h);
            }
            check(rc);
        });
    aggregator.def("merge", &PyAgg::merge)
        .def("get_result", &PyAgg::get_result)
        .def("__sizeof__", &PyAgg::bytes_used)
        .def("set_data", &PyAgg::set_data, py::arg("thread"), py::arg("ar"), py::arg("index") = 0)
        .def("clear_data_mask", &PyAgg::clear_data_mask)
        .def("set_data_mask", &PyAgg::set_data_mask)
        .def("set_selection", &PyAgg::set_selection)
        .def("device_touch", &PyAgg::device_touch)
        .def("reset", &PyAgg::reset)
        .def_property_readonly("__cuda_array_interface__", &PyAgg::device_grid_interface)
        .def_property_readonly("grid", [](const PyAgg &a) { return a.grid; }, py::return_value_policy::reference);

    py::class_<PyBinner> binner(m, "Binner");
    binner.def("set_data", &PyBinner::set_data)
        .def("clear_data_mask", &PyBinner::clear_data_mask)
        .def("set_data_mask", &PyBinner::set_data_mask)
        .def("__len__", &PyBinner::shape)
        .def_property_readonly("expression", [](const PyBinner &b) { return b.expression; });

    py::class_<PyBinnerScalar, PyBinner> scalar_base(m, "BinnerScalar");
    scalar_base.def_property_readonly("bins", [](const PyBinnerScalar &b) { return b.bins; })
        .def_property_readonly("vmin", [](const PyBinnerScalar &b) { return b.vmin; })
        .def_property_readonly("vmax", [](const PyBinnerScalar &b) { return b.vmax; })
        .def("set_float32_scaling", [](PyBinnerScalar &b, int mode) { check(vxh_binner_scalar_set_f32_scaling(b.h, mode)); }); // (the legacy statisticNd_f4 arithmetic)
    py::class_<PyBinnerOrdinal, PyBinner> ordinal_base(m, "BinnerOrdinal");
    ordinal_base.def_property_readonly("ordinal_count", [](const PyBinnerOrdinal &b) { return b.ordinal_count; })
        .def_property_readonly("min_value", [](const PyBinnerOrdinal &b) { return b.min_value; })
        .def_property_readonly("allow_other", [](const PyBinnerOrdinal &b) { return b.allow_other; })
        .def_property_readonly("invert", [](const PyBinnerOrdinal &b) { return b.invert; });
    py::class_<PyBinnerHash, PyBinner> hash_base(m, "BinnerHash");
    hash_base.def_property_readonly("hash_bins", [](const PyBinnerHash &b) { return (int64_t)b.shape() - 2; });

    py::class_<PyGrid>(m, "Grid")
        .def(py::init<std::vector<PyBinner *>>(), py::keep_alive<1, 2>())
        .def("bin", [](PyGrid &g, int thread, const py::list &aggs, py::object length) {
            // AggFirst aggregators take their own passes (vxh_first_bin), everything else ONE fused vxh_grid_bin
            std::vector<PyAgg *> plain;
            std::vector<PyAggFirst *> firsts;
            std::vector<PyCollect *> collects;
            for (const py::handle &a : aggs) {
                if (py::isinstance<PyAggFirst>(a)) firsts.push_back(a.cast<PyAggFirst *>());
                else if (py::isinstance<PyCollect>(a)) collects.push_back(a.cast<PyCollect *>());
                else plain.push_back(a.cast<PyAgg *>());
            }
            uint64_t n;
            if (length.is_none()) {
                if (g.binners.empty()) throw std::runtime_error("no binners set and no length given"); // src/agg.hpp:78
                n = vxh_binner_data_length(g.binners[0]->h, thread);
            } else {
                n = length.cast<uint64_t>();
            }
            if (!plain.empty()) g.bin(thread, plain, n);
            for (PyAggFirst *f : firsts) {
                int rc;
                {
                    py::gil_scoped_release release;
                    rc = vxh_first_bin(f->h, thread, n);
                }
                check(rc);
            }
            for (PyCollect *c : collects) c->bin(thread, n);
        }, py::arg("thread"), py::arg("aggregators"), py::arg("length") = py::none())
        .def("__len__", [](const PyGrid &g) { return vxh_grid_length1d(g.h); })
        .def_property_readonly("binners", [](const PyGrid &g) { return g.binners; }, py::return_value_policy::reference)
        .def_property_readonly("shapes", &PyGrid::shapes)
        .def_property_readonly("strides", &PyGrid::strides);

    py::class_<PyAggFirst>(m, "AggFirst")
        .def(py::init<PyGrid *, int, int, bool, int, int, bool>(), py::keep_alive<1, 2>(), py::arg("grid"), py::arg("grids"), py::arg("threads"), py::arg("invert"),
             py::arg("dtype"), py::arg("dtype_order"), py::arg("flip") = false)
        .def("set_data", &PyAggFirst::set_data, py::arg("thread"), py::arg("ar"), py::arg("index") = 0)
        .def("set_data_mask", &PyAggFirst::set_data_mask)
        .def("clear_data_mask", &PyAggFirst::clear_data_mask)
        .def("get_result", &PyAggFirst::get_result)
        .def("raw_result", &PyAggFirst::raw_result)
        .def("merge", [](PyAggFirst &, const py::object &) { throw std::runtime_error("merge: not implemented"); }) // src/agg_first.cpp:42
        .def("__sizeof__", &PyAggFirst::bytes_used)
        .def_property_readonly("grid", [](const PyAggFirst &a) { return a.grid; }, py::return_value_policy::reference);
    py::class_<PyCollect>(m, "AggCollect")
        .def(py::init<PyGrid *, int, int, bool, bool, int, int, bool>(), py::keep_alive<1, 2>(), py::arg("grid"), py::arg("grids"), py::arg("threads"), py::arg("a"), py::arg("b"),
             py::arg("mode"), py::arg("dtype"), py::arg("flip") = false)
        .def("pairs", &PyCollect::pairs)
        .def("merge_pairs", &PyCollect::merge_pairs)
        .def("set_data", &PyCollect::set_data, py::arg("thread"), py::arg("ar"), py::arg("index") = 0)
        .def("set_data_mask", &PyCollect::set_data_mask)
        .def("clear_data_mask", &PyCollect::clear_data_mask)
        .def("set_selection_mask", &PyCollect::set_selection_mask)
        .def("clear_selection_mask", &PyCollect::clear_selection_mask)
        .def("bin", &PyCollect::bin)
        .def("get_result", &PyCollect::get_result)
        .def("list_arrays", &PyCollect::list_arrays)
        .def("merge", [](PyCollect &c, const py::object &others) {
            if (c.mode == 0 && py::len(others) > 0) throw std::runtime_error("merge not implemented"); // src/agg_nunique.cpp:46-49 (AggList: a no-op, src/agg_list.cpp:49)
        })
        // (vaex predicts nunique's footprint as sizeof(the class on a grid of one cell) x cells and insists on equality:
        //  vaex/agg.py:354-368 — any per-cell constant does; the pairs live on the device)
        .def("__sizeof__", [](const PyCollect &c) { return (size_t)8 * (size_t)vxh_grid_length1d(c.grid->h); })
        .def_property_readonly("grid", [](const PyCollect &a) { return a.grid; }, py::return_value_policy::reference);
    // AggFirst_<T>_<T2>[_non_native] -> a constructor with the reference's signature (grid, grids, threads, invert)
    m.def("__getattr__", [m](const std::string &name) -> py::object {
        for (int mode = 0; mode < 2; mode++) {
            const std::string prefix = mode ? "AggList_" : "AggNUnique_";
            if (name.compare(0, prefix.size(), prefix) != 0) continue;
            std::string rest = name.substr(prefix.size());
            bool flip = false;
            const std::string nn = "_non_native";
            if (rest.size() > nn.size() && rest.compare(rest.size() - nn.size(), nn.size(), nn) == 0) {
                flip = true;
                rest = rest.substr(0, rest.size() - nn.size());
            }
            for (int a = 0; a < VXH_DTYPE_COUNT; a++) {
                bool hit = rest == kTypeNames[a];
                if (mode) { // AggList_<T>_<T2>: the second type is registered and never used
                    hit = false;
                    for (int b = 0; b < VXH_DTYPE_COUNT; b++) hit = hit || rest == std::string(kTypeNames[a]) + "_" + kTypeNames[b];
                }
                if (hit) return py::module::import("functools").attr("partial")(m.attr("AggCollect"), py::arg("mode") = mode, py::arg("dtype") = a, py::arg("flip") = flip);
            }
        }
        const std::string prefix = "AggFirst_";
        if (name.compare(0, prefix.size(), prefix) == 0) {
            std::string rest = name.substr(prefix.size());
            bool flip = false;
            const std::string nn = "_non_native";
            if (rest.size() > nn.size() && rest.compare(rest.size() - nn.size(), nn.size(), nn) == 0) {
                flip = true;
                rest = rest.substr(0, rest.size() - nn.size());
            }
            for (int a = 0; a < VXH_DTYPE_COUNT; a++)
                for (int b = 0; b < VXH_DTYPE_COUNT; b++)
                    if (rest == std::string(kTypeNames[a]) + "_" + kTypeNames[b]) {
                        py::object cls = m.attr("AggFirst");
                        return py::module::import("functools").attr("partial")(cls, py::arg("dtype") = a, py::arg("dtype_order") = b, py::arg("flip") = flip);
                    }
        }
        throw py::attribute_error("module 'vaex_amd.superagg' has no attribute '" + name + "'");
    });

    py::class_<PyHashMap> hashmap(m, "ordered_set");
    hashmap.def("update", &PyHashMap::update, py::arg("keys"), py::arg("mask") = py::none())
        .def("map_ordinal", &PyHashMap::map_ordinal)
        .def("map_ordinal_device", [](PyHashMap &h, const py::object &keys) {
            ArrayRef k = resolve_array(keys);
            auto out = std::make_unique<PyDeviceArray>(k.n, VXH_I64);
            h.map_ordinal_device(keys, (uintptr_t)out->p, out->n);
            return out;
        })
        .def("set_keys", &PyHashMap::set_keys)
        .def("key_array", &PyHashMap::key_array)
        .def("__len__", &PyHashMap::count)
        .def_property_readonly("null_index", &PyHashMap::null_index)
        .def_property_readonly("has_null", [](PyHashMap &h) { return h.null_index() >= 0; });

    add_type<VXH_F64>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_F32>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_I64>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_I32>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_I16>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_I8>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_U64>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_U32>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_U16>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_U8>(m, scalar_base, ordinal_base, aggregator);
    add_type<VXH_BOOL>(m, scalar_base, ordinal_base, aggregator);

    add_binner_hash<VXH_F64, false>(m, hash_base);
    add_binner_hash<VXH_F64, true>(m, hash_base);
    add_binner_hash<VXH_F32, false>(m, hash_base);
    add_binner_hash<VXH_F32, true>(m, hash_base);
    add_binner_hash<VXH_I64, false>(m, hash_base);
    add_binner_hash<VXH_I64, true>(m, hash_base);
    add_binner_hash<VXH_I32, false>(m, hash_base);
    add_binner_hash<VXH_I32, true>(m, hash_base);
    add_binner_hash<VXH_I16, false>(m, hash_base);
    add_binner_hash<VXH_I16, true>(m, hash_base);
    add_binner_hash<VXH_I8, false>(m, hash_base);
    add_binner_hash<VXH_I8, true>(m, hash_base);
    add_binner_hash<VXH_U64, false>(m, hash_base);
    add_binner_hash<VXH_U64, true>(m, hash_base);
    add_binner_hash<VXH_U32, false>(m, hash_base);
    add_binner_hash<VXH_U32, true>(m, hash_base);
    add_binner_hash<VXH_U16, false>(m, hash_base);
    add_binner_hash<VXH_U16, true>(m, hash_base);
    add_binner_hash<VXH_U8, false>(m, hash_base);
    add_binner_hash<VXH_U8, true>(m, hash_base);
    add_binner_hash<VXH_BOOL, false>(m, hash_base);
    add_binner_hash<VXH_BOOL, true>(m, hash_base);
    add_hash<VXH_I64>(m, hashmap);
    add_hash<VXH_I32>(m, hashmap);
    add_hash<VXH_I16>(m, hashmap);
    add_hash<VXH_I8>(m, hashmap);
    add_hash<VXH_U64>(m, hashmap);
    add_hash<VXH_U32>(m, hashmap);
    add_hash<VXH_U16>(m, hashmap);
    add_hash<VXH_U8>(m, hashmap);
    add_hash<VXH_BOOL>(m, hashmap);
}
