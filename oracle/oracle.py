"""TEST INFRASTRUCTURE — ctypes wrapper around oracle/vaex_oracle.c (the CPU restatement of
vaex's binned-statistics algorithm) plus a loader for oracle/_ref (the reference's own C++,
compiled in place by oracle/build_ref.sh).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (vaex_amd/) never does.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libvaex_oracle.so")
SRC = os.path.join(HERE, "vaex_oracle.c")

DTYPES = ["float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"]
DT = {name: i for i, name in enumerate(DTYPES)}
KINDS = {"count": 0, "sum": 1, "summoment": 2, "min": 3, "max": 4}


def build(force=False):
    """gcc -O2 build of the restatement (no FMA contraction: the bin index must round like the reference)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, SRC, "-lm"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        vp, u8p, u64, i64, dbl, i32 = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int64, ctypes.c_double, ctypes.c_int
        L.vxo_bin_scalar.argtypes = [i32, i32, vp, u8p, u64, dbl, dbl, u64, u64, vp]
        L.vxo_bin_ordinal.argtypes = [i32, i32, vp, u8p, u64, i64, i64, i32, i32, u64, vp]
        L.vxo_agg_count.argtypes = [i32, i32, vp, u8p, vp, u64, vp]
        L.vxo_agg_sum.argtypes = [i32, i32, vp, u8p, vp, u64, vp, i32, ctypes.c_uint32]
        L.vxo_agg_minmax.argtypes = [i32, i32, vp, u8p, vp, u64, vp, i32]
        L.vxo_minmax_fill.argtypes = [i32, i32, vp]
        L.vxo_upcast_kind.argtypes = [i32]
        L.vxo_set_create.argtypes = [u64]
        L.vxo_set_create.restype = ctypes.c_void_p
        L.vxo_set_free.argtypes = [vp]
        L.vxo_set_count.argtypes = [vp]
        L.vxo_set_count.restype = i64
        L.vxo_set_update.argtypes = [vp, vp, u64]
        L.vxo_set_map_ordinal.argtypes = [vp, vp, u64, vp]
        L.vxo_set_keys.argtypes = [vp, vp]
        L.vxo_statistic_nd.argtypes = [vp, i32, vp, u64, vp, vp, vp, vp, i32, i32, i32, vp]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def dtype_code(arr_or_name):
    """(dtype code, flip) of a numpy array / dtype name; big-endian arrays are `_non_native`."""
    if isinstance(arr_or_name, str):
        return DT[arr_or_name], 0
    dt = arr_or_name.dtype
    flip = 0 if dt.byteorder in "<=|" or (dt.byteorder == "=" and sys.byteorder == "little") else 1
    if dt.byteorder == ">" and sys.byteorder == "little":
        flip = 1
    name = dt.newbyteorder("=").name
    if dt.kind in "mM":
        name = "int64"
    return DT[name], flip


def _mask_u8(m):
    if m is None:
        return None
    return np.ascontiguousarray(m).view(np.uint8) if m.dtype == np.bool_ else np.ascontiguousarray(m, dtype=np.uint8)


def binner_shape(b):
    if b["kind"] == "scalar":
        return int(b["bins"]) + 3
    return int(b["count"]) + (3 if b.get("allow_other") else 2)


def flat_indices(binners, n):
    """Sum over dims of sub_index * stride (dim 0 stride 1) — src/agg.hpp:63-73, :106-137."""
    L = lib()
    idx = np.zeros(n, dtype=np.uint64)
    stride = 1
    for b in binners:
        data = np.ascontiguousarray(b["data"][:n])
        mask = _mask_u8(b.get("mask"))
        if mask is not None:
            mask = np.ascontiguousarray(mask[:n])
        code, flip = dtype_code(data)
        if b["kind"] == "scalar":
            L.vxo_bin_scalar(code, flip, _ptr(data), _ptr(mask), n, float(b["vmin"]), float(b["vmax"]), int(b["bins"]), stride, _ptr(idx))
        else:
            L.vxo_bin_ordinal(code, flip, _ptr(data), _ptr(mask), n, int(b["count"]), int(b.get("min_value", 0)), int(bool(b.get("allow_other"))), int(bool(b.get("invert"))), stride, _ptr(idx))
        stride *= binner_shape(b)
    return idx


def grid_dtype(kind, data_dtype_name):
    """numpy dtype of one grid cell, as the reference exposes it."""
    if kind == "count":
        return np.dtype("int64")
    if kind in ("sum", "summoment"):
        k = lib().vxo_upcast_kind(DT[data_dtype_name])
        return np.dtype(["float64", "int64", "uint64"][k])
    return np.dtype(data_dtype_name)


def aggregate(agg, idx, n, cells, grid=None):
    """One aggregator over precomputed flat indices; returns the 1-d grid (length cells)."""
    L = lib()
    kind = agg["kind"]
    data = agg.get("data")
    mask = _mask_u8(agg.get("mask"))
    if mask is not None:
        mask = np.ascontiguousarray(mask[:n])
    if data is not None:
        data = np.ascontiguousarray(data[:n])
        code, flip = dtype_code(data)
        name = DTYPES[code]
    else:
        code, flip, name = DT[agg.get("dtype", "int64")], 0, agg.get("dtype", "int64")
    gdt = grid_dtype(kind, name)
    if grid is None:
        grid = np.zeros(cells, dtype=gdt)
        if kind in ("min", "max"):
            elem = np.zeros(1, dtype=gdt)
            L.vxo_minmax_fill(code, int(kind == "max"), _ptr(elem))
            grid[:] = elem[0]
    if kind == "count":
        L.vxo_agg_count(code, flip, _ptr(data), _ptr(mask), _ptr(idx), n, _ptr(grid))
    elif kind == "sum":
        L.vxo_agg_sum(code, flip, _ptr(data), _ptr(mask), _ptr(idx), n, _ptr(grid), 0, 0)
    elif kind == "summoment":
        L.vxo_agg_sum(code, flip, _ptr(data), _ptr(mask), _ptr(idx), n, _ptr(grid), 1, int(agg.get("moment", 2)))
    elif kind in ("min", "max"):
        L.vxo_agg_minmax(code, flip, _ptr(data), _ptr(mask), _ptr(idx), n, _ptr(grid), int(kind == "max"))
    else:
        raise ValueError(kind)
    return grid


def run_case(case):
    """All aggregators of a case -> list of N-d result grids (shape = binner shapes, result[i0, i1, ...])."""
    n = int(case["n"])
    binners = case["binners"]
    shapes = [binner_shape(b) for b in binners]
    cells = int(np.prod(shapes)) if shapes else 1
    idx = flat_indices(binners, n)
    out = []
    for agg in case["aggs"]:
        g = aggregate(agg, idx, n, cells)
        out.append(g.reshape(shapes[::-1]).T if shapes else g.reshape(()))
    return out


class OrderedSet:
    """ordered_set<int64> restatement (first-seen ordinals)."""

    def __init__(self, capacity=1 << 16):
        cap = 1
        while cap < capacity * 2:
            cap <<= 1
        self._h = lib().vxo_set_create(cap)

    def __del__(self):
        try:
            lib().vxo_set_free(self._h)
        except Exception:
            pass

    def update(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        lib().vxo_set_update(self._h, _ptr(keys), len(keys))

    def __len__(self):
        return int(lib().vxo_set_count(self._h))

    def map_ordinal(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty(len(keys), dtype=np.int64)
        lib().vxo_set_map_ordinal(self._h, _ptr(keys), len(keys), _ptr(out))
        return out

    def key_array(self):
        out = np.empty(len(self), dtype=np.int64)
        lib().vxo_set_keys(self._h, _ptr(out))
        return out


def minmax(data, mask=None):
    """Legacy statisticNd OP_MIN_MAX on a 0-d grid (plain < / >; NaN never wins)."""
    d = np.asarray(data).astype(np.float64)
    if mask is not None:
        d = d[np.asarray(mask).astype(bool)]
    d = d[d == d]
    if len(d) == 0:
        return np.inf, -np.inf
    return float(d.min()), float(d.max())


STAT_FIELDS = {0: 1, 1: 1, 2: 2, 3: 2, 4: 3}


def statistic_nd(blocks, weights, grid, minima, maxima, op, use_edges=0):
    """vaexfast.statisticNd_f8 restated (vxo_statistic_nd): accumulates into the float64 `grid` of shape
    sizes + (fields,), same argument order as vaexfast.cpp:1361-1370."""
    blocks = [np.ascontiguousarray(b, dtype=np.float64) for b in blocks]
    nd = len(blocks)
    assert grid.dtype == np.float64 and grid.ndim == nd + 1 and grid.flags.c_contiguous
    fields = grid.shape[-1]
    w = None
    if weights is not None:
        w = np.ascontiguousarray(weights[0] if isinstance(weights, (list, tuple)) else weights, dtype=np.float64)
    n = len(blocks[0]) if blocks else (len(w) if w is not None else 0)
    ptrs = (ctypes.c_void_p * max(1, nd))(*[b.ctypes.data for b in blocks])
    mins = np.asarray(minima, dtype=np.float64)
    maxs = np.asarray(maxima, dtype=np.float64)
    counts = np.asarray(grid.shape[:-1], dtype=np.int64)
    strides = np.asarray([s // (8 * fields) for s in grid.strides[:-1]], dtype=np.int64)
    lib().vxo_statistic_nd(ptrs, nd, _ptr(w), n, _ptr(mins), _ptr(maxs), _ptr(counts), _ptr(strides), op, fields, int(use_edges), _ptr(grid))


# ------------------------------------------------------------------------------------------
# the reference's own compiled C++ (oracle/_ref), when present
# ------------------------------------------------------------------------------------------
_ref = {}


def ref_module(name="superagg"):
    """Import oracle/_ref/<name>*.so (built by oracle/build_ref.sh); None when it is not there."""
    if name in _ref:
        return _ref[name]
    import glob
    import importlib.util
    mod = None
    hits = glob.glob(os.path.join(HERE, "_ref", name + ".*.so"))
    if hits:
        try:
            spec = importlib.util.spec_from_file_location(name, hits[0])
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        except Exception:  # pragma: no cover - e.g. ABI mismatch on another box
            mod = None
    _ref[name] = mod
    return mod
