"""Test helpers: seeded cases for the binned-statistics path and a runner that drives ANY module with
the `vaex.superagg` class surface (the product `vaex_amd.superagg`, or the reference's own C++ in
oracle/_ref) through the exact call sequence vaex's TaskPartAggregation.process uses
(/root/reference/packages/vaex-core/vaex/cpu.py:678-786): set_data / set_data_mask /
clear_data_mask on every binner and aggregator for a thread slot, then Grid.bin(thread, aggs, N)."""
import sys

import numpy as np

from oracle import oracle

KIND_CLASS = {"count": "AggCount_", "sum": "AggSum_", "summoment": "AggSumMoment_", "min": "AggMin_", "max": "AggMax_"}


def class_postfix(arr_or_name):
    if isinstance(arr_or_name, str):
        return arr_or_name
    code, flip = oracle.dtype_code(arr_or_name)
    return oracle.DTYPES[code] + ("_non_native" if flip else "")


def _as_u8_mask(m):
    return np.ascontiguousarray(m).view(np.uint8) if m.dtype == np.bool_ else np.ascontiguousarray(m, dtype=np.uint8)


def _dev(x, to_device):
    if x is None or to_device is None:
        return x
    return to_device(x)


def run_superagg(sa, case, nthreads=1, chunk=None, grids=None, to_device=None, keep=None):
    """Run `case` through module `sa`; returns the list of get_result() arrays (one per aggregator).

    chunk: rows per Grid.bin call (None = everything at once); chunks are dealt round-robin to
    `nthreads` slots, like the executor's thread pool does.  to_device: optional callable turning a
    numpy chunk into a device array (HBM-resident path).  keep: optional list collecting the
    aggregator objects (to poke at buffers afterwards)."""
    n = int(case["n"])
    binners = []
    for b in case["binners"]:
        pf = class_postfix(b["data"])
        if b["kind"] == "scalar":
            binners.append(getattr(sa, "BinnerScalar_" + pf)(nthreads, b.get("expression", "x"), float(b["vmin"]), float(b["vmax"]), int(b["bins"])))
        else:
            binners.append(getattr(sa, "BinnerOrdinal_" + pf)(nthreads, b.get("expression", "k"), int(b["count"]), int(b.get("min_value", 0)), bool(b.get("allow_other", False)), bool(b.get("invert", False))))
    grid = sa.Grid(binners)
    aggs = []
    for a in case["aggs"]:
        pf = class_postfix(a["data"]) if a.get("data") is not None else a.get("dtype", "int64")
        cls = getattr(sa, KIND_CLASS[a["kind"]] + pf)
        ng = grids or nthreads
        if a["kind"] == "summoment":
            aggs.append(cls(grid, ng, nthreads, int(a.get("moment", 2))))
        else:
            aggs.append(cls(grid, ng, nthreads))
    if keep is not None:
        keep.extend(aggs)
        keep.append(grid)
    chunk = chunk or max(n, 1)
    refs = []
    t = 0
    for i1 in range(0, n, chunk):
        i2 = min(n, i1 + chunk)
        for bobj, b in zip(binners, case["binners"]):
            d = _dev(np.ascontiguousarray(b["data"][i1:i2]), to_device)
            refs.append(d)
            bobj.set_data(t, d)
            if b.get("mask") is not None:
                m = _dev(_as_u8_mask(b["mask"][i1:i2]), to_device)
                refs.append(m)
                bobj.set_data_mask(t, m)
            else:
                bobj.clear_data_mask(t)
        for aobj, a in zip(aggs, case["aggs"]):
            if a.get("data") is not None:
                d = _dev(np.ascontiguousarray(a["data"][i1:i2]), to_device)
                refs.append(d)
                aobj.set_data(t, d, 0)
            if a.get("mask") is not None:
                m = _dev(_as_u8_mask(a["mask"][i1:i2]), to_device)
                refs.append(m)
                aobj.set_data_mask(t, m)
            else:
                aobj.clear_data_mask(t)
        grid.bin(t, aggs, i2 - i1)
        t = (t + 1) % nthreads
    return [np.array(a.get_result()) for a in aggs]


def torch_device_array(x):
    """numpy -> cuda tensor exposing __cuda_array_interface__ (big-endian / bool arrays go as same-width ints)."""
    import torch
    x = np.ascontiguousarray(x)
    if x.dtype.byteorder == ">" or (x.dtype.byteorder == "=" and sys.byteorder == "big"):
        x = x.view(x.dtype.newbyteorder("<"))  # raw bytes unchanged
    if x.dtype == np.bool_:
        x = x.view(np.uint8)
    if x.dtype in (np.uint16, np.uint32, np.uint64):
        x = x.view({2: np.int16, 4: np.int32, 8: np.int64}[x.dtype.itemsize])
    return torch.from_numpy(x).cuda()


# ------------------------------------------------------------------------------------------
# seeded inputs: x,y,z ~ N(0,1), v ~ N(3,2), limits [-4,4], 1e-4 NaNs, masks
# ------------------------------------------------------------------------------------------
def gaussian_columns(n, seed=42, nan_fraction=1e-4):
    rng = np.random.default_rng(seed)
    cols = {name: rng.normal(0, 1, n) for name in "xyz"}
    cols["v"] = rng.normal(3, 2, n)
    if nan_fraction:
        for name in ("x", "v"):
            k = max(1, int(n * nan_fraction))
            cols[name][rng.integers(0, n, k)] = np.nan
    cols["x"][: min(n, 3)] = [-np.inf, np.inf, 4.0][: min(n, 3)]  # +-inf and vmax itself -> edge cells
    return cols


def case_2d_count_mean(n, shape=256, seed=42, selection=False):
    c = gaussian_columns(n, seed)
    aggs = [dict(kind="count"), dict(kind="sum", data=c["v"]), dict(kind="count", data=c["v"])]
    if selection:
        m = c["v"] > 3
        for a in aggs:
            a["mask"] = m
    return dict(n=n, binners=[dict(kind="scalar", data=c["x"], vmin=-4, vmax=4, bins=shape), dict(kind="scalar", data=c["y"], vmin=-4, vmax=4, bins=shape)], aggs=aggs)


def case_3d_selection(n, shape=128, seed=42):
    c = gaussian_columns(n, seed)
    m = c["v"] > 3
    return dict(n=n, binners=[dict(kind="scalar", data=c[k], vmin=-4, vmax=4, bins=shape) for k in "xyz"], aggs=[dict(kind="count", mask=m)])


def case_groupby(n, groups=1000, seed=42, key_dtype="int64"):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, groups, n).astype(key_dtype)
    v = rng.normal(3, 2, n)
    v[rng.integers(0, n, max(1, n // 10000))] = np.nan
    return dict(n=n, binners=[dict(kind="ordinal", data=k, count=groups, min_value=0)],
                aggs=[dict(kind="sum", data=v), dict(kind="count", data=v), dict(kind="summoment", data=v, moment=2), dict(kind="min", data=v), dict(kind="max", data=v)])


def assert_case_equal(got, want, case, rtol=1e-12):
    """bit-exact for integer grids; |got-want| <= rtol * sum|v| per cell for float sums (SURVEY §7:
    the accumulation order differs, so the bound is relative to the magnitude summed into the cell)."""
    assert len(got) == len(want)
    for g, w, a in zip(got, want, case["aggs"]):
        assert g.shape == w.shape, (g.shape, w.shape)
        assert g.dtype == w.dtype, (g.dtype, w.dtype)
        if g.dtype.kind in "iub" or a["kind"] in ("min", "max", "count"):
            np.testing.assert_array_equal(g, w)
        else:
            n = int(case["n"])
            idx = oracle.flat_indices(case["binners"], n)
            data = np.asarray(a["data"][:n]).astype(np.float64)
            mag = np.abs(data) ** (a.get("moment", 2) if a["kind"] == "summoment" else 1)
            keep = data == data
            if a.get("mask") is not None:
                keep &= np.asarray(a["mask"][:n]).astype(bool)
            scale = np.bincount(idx[keep].astype(np.int64), weights=mag[keep], minlength=w.size)
            scale = scale.reshape(w.shape[::-1]).T if w.ndim else scale.reshape(())
            err = np.abs(g - w)
            assert np.all(err <= rtol * scale + 0.0), float(np.max(err / np.maximum(scale, 1e-300)))
