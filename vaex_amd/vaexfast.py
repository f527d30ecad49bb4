"""Legacy `vaex.vaexfast.statisticNd_f8` on the GPU.

vaexfast.cpp:1361-1510 is the Python entry, :1168-1278 the row loop, :1062-1164 the per-cell ops.  The legacy task
(vaex/cpu.py:510-610, TaskPartStatistic) calls
    statisticNd_f8(blocks, weights, grid, minima, maxima, op_code, use_edges)
once per chunk with `grid` a float64 array of shape (size_0, ..., size_{d-1}, fields) that it accumulates INTO.

Nothing here computes on the host: every op is a fused pass of the same HIP kernels that serve vaex.superagg
(BinnerScalar cells are a superset of statisticNd's: the interior cells use the identical expression
`(int)(scaled * bins)`, vaexfast.cpp:1227-1229 vs agg.hpp BinnerScalar; use_edges=1 IS the BinnerScalar layout,
vaexfast.cpp:1189-1209), followed by a cell-wise fold into the caller's grid:

    OP_ADD1 (0)                 grid[...,0] += count(*)
    OP_COUNT (1)                grid[...,0] += count(weights[0])            (NaN weights skipped, :1069-1075)
    OP_MIN_MAX (2)              grid[...,0] = min(grid[...,0], min(w)); grid[...,1] = max(..)   (:1090-1101)
    OP_ADD_WEIGHT_MOMENTS_01 (3)   count, sum
    OP_ADD_WEIGHT_MOMENTS_012 (4)  count, sum, sum of squares
    OP_COV (5)                  per cell [count_c (N), sum_c (N), pair counts (N x N), pair sums of products (N x N)] of the N
                                weight columns, a pair counted where both are not NaN (:1117-1153; df.cov, vaex/dataframe.py:1456):
                                count / sum / sum-of-squares aggregators per column, count / sum aggregators over the
                                row-wise product column (vxh_product_f64) per pair.  One deviation, stated: a pair inf x 0
                                (product NaN, neither input NaN) is skipped where the reference adds NaN to the cell.
    OP_FIRST (6)                grid[...,0] = weights[0] of the row with the smallest weights[1] seen so far, grid[...,1] that
                                order value (:1155-1166; vaex/dataframe.py:975): the AggFirst passes (vxh_first_*) over the value
                                column's BIT PATTERN (op_first compares the order only: a NaN value can win, which AggFirst
                                would skip), folded into the caller's grid with the same `order < grid[...,1]` rule.
`statisticNd_f4` (round 3) is the same over float32 blocks / weights with the reference's float32 ARITHMETIC: `T scales[d] = 1 /
(maxima[d] - minima[d])` and `(value - minima[d]) * scales[d]` are float32 operations (vaexfast.cpp:1185-1190), the product with the
bin count is a double one — except in the two-dimensional loop, whose `T scaled` keeps it in float32 (:1240-1246).  The scalar
binner takes that arithmetic through vxh_binner_scalar_set_f32_scaling (mode 2 for two dimensions without edges, else 1); the ops
accumulate in double exactly like the float64 entry.  OP_COV over float32 weights is not offered (NotImplementedError: install()
keeps the reference's function for it)."""
import threading

import numpy as np

from . import superagg as _sa

# the entry has no thread-slot argument (the legacy task calls it from every pool thread with private grids): the
# calls share slot 0 of the library, one at a time
_LOCK = threading.Lock()

OP_ADD1, OP_COUNT, OP_MIN_MAX, OP_ADD_WEIGHT_MOMENTS_01, OP_ADD_WEIGHT_MOMENTS_012, OP_COV, OP_FIRST = range(7)
_FIELDS = {OP_ADD1: 1, OP_COUNT: 1, OP_MIN_MAX: 2, OP_ADD_WEIGHT_MOMENTS_01: 2, OP_ADD_WEIGHT_MOMENTS_012: 3, OP_FIRST: 2}


def _is_device(a):
    return hasattr(a, "__cuda_array_interface__") and not isinstance(a, np.ndarray)


def _f8(a, what, size=8):
    name = f"statisticNd_f{size}"
    if _is_device(a):
        if a.__cuda_array_interface__["typestr"] not in (f"<f{size}", f">f{size}"):
            raise TypeError(f"{name}: {what} must be float{size * 8}")
        return a
    a = np.asarray(a)
    if a.dtype.kind != "f" or a.dtype.itemsize != size:
        raise TypeError(f"{name}: {what} must be float{size * 8}, not {a.dtype}")
    if a.ndim != 1:
        raise ValueError(f"{name}: {what} must be 1-dimensional")
    return np.ascontiguousarray(a)


def _postfix(a):
    if _is_device(a):
        t = a.__cuda_array_interface__["typestr"]
        return ("float64" if t[1:] == "f8" else "float32") + ("" if t[0] == "<" else "_non_native")
    return ("float64" if a.dtype.itemsize == 8 else "float32") + ("" if a.dtype.isnative else "_non_native")


#: The thread slot every call of this module bins on.  vaex calls statisticNd from its pool threads DURING a pass (TaskPartStatistic.process,
#: vaex/cpu.py:600-611) — possibly the same pass in which the aggregation task parts of other delayed calls bin on the pool's own slots
#: 0 .. nthreads-1 (df.minmax(delay=True) next to df.count(binby=..., delay=True)).  A slot must only be driven by one host thread at a time
#: (include/vaex_hip.h), so this entry stays off the pool's slots: it serialises itself (_LOCK) on the first of the library's auxiliary slots.
#: (Until round 5 it used slot 0: a delayed minmax in the same pass as a binned aggregation corrupted one of the two now and then — found by
#: tests/test_vaex_random_calls.py's soak run, 7 of 8000 calls.)
_SLOT = int(_sa.AUX_SLOT)
_T = _SLOT + 1


def statisticNd_f8(blocks, weights, grid, minima, maxima, op_code, use_edges=0):
    """Accumulates one chunk into `grid` (in place); returns None like the reference."""
    with _LOCK:
        return _statistic_nd(blocks, weights, grid, minima, maxima, op_code, use_edges)


def statisticNd_f4(blocks, weights, grid, minima, maxima, op_code, use_edges=0):
    """The float32 entry (vaexfast.cpp:1361-1510 with T = float): float32 blocks and weights, the float32 scaling arithmetic of the
    reference's loops, a float64 grid."""
    with _LOCK:
        return _statistic_nd(blocks, weights, grid, minima, maxima, op_code, use_edges, size=4)


def _statistic_nd(blocks, weights, grid, minima, maxima, op_code, use_edges, size=8):
    if not isinstance(blocks, (list, tuple)):
        raise ValueError("statisticNd_: blocklist (first argument) is not a list")
    if op_code not in _FIELDS and op_code != OP_COV:
        raise ValueError(f"statisticNd_wrap_template_endian: unknown op code {op_code} for statistic")
    blocks = [_f8(b, "block", size) for b in blocks]
    nd = len(blocks)
    if weights is None:
        wlist = []
    elif isinstance(weights, (list, tuple)):
        wlist = [_f8(w, "weight", size) for w in weights]
    else:
        wlist = [_f8(weights, "weight", size)]
    if op_code != OP_ADD1 and not wlist:
        raise ValueError("statisticNd_: this op needs a weight array")
    if not isinstance(grid, np.ndarray) or grid.dtype != np.float64:
        raise TypeError("statisticNd_: grid must be a float64 ndarray")
    if grid.ndim > nd + 1:
        # The reference never compares the grid's rank with the blocks': it takes the first `nd` sizes / strides as the binned dimensions and
        # writes field f of a cell at the cell's offset + f (vaexfast.cpp:197-213, :1485-1493) — i.e. everything behind the binned dimensions is
        # ONE run of values per cell (the reference's own unittest hands a (10, 2) grid to a call without blocks: vaex/test/cmodule.py:26-35)
        flat = grid.view()
        try:
            flat.shape = grid.shape[:nd] + (-1,)
        except AttributeError:
            raise ValueError("statisticNd_: the grid's trailing dimensions are not contiguous")
        grid = flat
    if grid.ndim != nd + 1:
        raise ValueError(f"statisticNd_: grid has {grid.ndim} dimensions, expected {nd + 1}")
    if grid.strides[-1] != 8:
        raise RuntimeError(f"last dimension in grid should have stride of 1, not {grid.strides[-1] // 8}")
    ncol = len(wlist)
    fields = 2 * ncol + 2 * ncol * ncol if op_code == OP_COV else _FIELDS[op_code]
    if op_code == OP_FIRST and ncol < 2:
        raise ValueError("statisticNd_: OP_FIRST needs a value and an order weight")
    if grid.shape[-1] < fields:
        raise ValueError(f"statisticNd_: op {op_code} writes {fields} values per cell, grid has {grid.shape[-1]}")
    if len(minima) != nd or len(maxima) != nd:
        raise ValueError("statisticNd_: minima/maxima must have one entry per block")
    lengths = {len(b) if not _is_device(b) else b.__cuda_array_interface__["shape"][0] for b in blocks + wlist}
    if len(lengths) > 1:
        raise ValueError("statisticNd_: blocks and weights differ in length")
    n = lengths.pop() if lengths else 0
    sizes = grid.shape[:-1]
    if use_edges and any(s < 4 for s in sizes):
        raise ValueError("statisticNd_: with edges every grid dimension needs at least 4 cells")

    binners = []
    for d, b in enumerate(blocks):
        bins = sizes[d] - 3 if use_edges else sizes[d]
        binner = getattr(_sa, "BinnerScalar_" + _postfix(b))(_T, f"block{d}", float(minima[d]), float(maxima[d]), int(bins))
        if size == 4:
            binner.set_float32_scaling(2 if (nd == 2 and not use_edges) else 1)
        binner.set_data(_SLOT, b)
        binners.append(binner)
    g = _sa.Grid(binners)
    inner = tuple(slice(None) if use_edges else slice(2, -1) for _ in range(nd))
    if op_code == OP_FIRST:
        w, order = wlist[0], wlist[1]
        if _postfix(w) != _postfix(order):
            raise NotImplementedError("statisticNd: OP_FIRST with weights of different byte order")
        # op_first looks at the ORDER only (:1160: a row whose value is NaN can win), AggFirst also skips NaN values
        # (src/agg_first.cpp:139): the value column goes in as its int64 bit pattern, which has no NaN
        if _is_device(w):
            import torch
            wbits = torch.as_tensor(w).view(torch.int64 if size == 8 else torch.int32)
        else:
            wbits = w.view(w.dtype.byteorder.replace("=", "<").replace("|", "<") + f"i{size}") if not w.dtype.isnative else w.view(f"i{size}")
        a = getattr(_sa, ("AggFirst_int64_" if size == 8 else "AggFirst_int32_") + _postfix(order))(g, 1, _T, False)
        a.set_data(_SLOT, wbits, 0)
        a.set_data(_SLOT, order, 1)
        if n:
            g.bin(_SLOT, [a], n)
        values, masked, orders = (np.asarray(r)[inner] for r in a.raw_result())
        values = values.view(f"f{size}")  # (same item size: fine for the strided and the 0-d result alike)
        take = ~masked & (orders < grid[..., 1])  # src/vaexfast.cpp:1160-1163
        grid[..., 0][take] = values[take]
        grid[..., 1][take] = orders[take]
        return None
    if op_code == OP_COV:
        if size != 8:
            raise NotImplementedError("statisticNd_f4: OP_COV")
        if any(_postfix(w) != "float64" for w in wlist):
            raise NotImplementedError("statisticNd: OP_COV on non-native weights")
        N = ncol
        per_col, per_pair = [], {}
        aggs = []
        for w in wlist:
            trio = [_sa.AggCount_float64(g, 1, _T), _sa.AggSum_float64(g, 1, _T), _sa.AggSumMoment_float64(g, 1, _T, 2)]
            for a in trio:
                a.set_data(_SLOT, w, 0)
            per_col.append(trio)
            aggs += trio
        keep = []
        for col in range(N):
            for row in range(col + 1, N):
                prod = _sa.product(wlist[col], wlist[row])
                duo = [_sa.AggCount_float64(g, 1, _T), _sa.AggSum_float64(g, 1, _T)]
                for a in duo:
                    a.set_data(_SLOT, prod, 0)
                per_pair[(col, row)] = duo
                aggs += duo
                keep.append(prod)
        if n:
            g.bin(_SLOT, aggs, n)
        res = lambda a: np.asarray(a.get_result())[inner]
        for col in range(N):
            cnt, sm, sq = (res(a) for a in per_col[col])
            grid[..., col] += cnt
            grid[..., col + N] += sm
            grid[..., 2 * N + col + col * N] += cnt
            grid[..., 2 * N + N * N + col + col * N] += sq
            for row in range(col + 1, N):
                pc, ps = (res(a) for a in per_pair[(col, row)])
                ia, ib = row + col * N, col + row * N
                grid[..., 2 * N + ia] += pc
                grid[..., 2 * N + ib] = grid[..., 2 * N + ia]
                grid[..., 2 * N + N * N + ia] += ps
                grid[..., 2 * N + N * N + ib] = grid[..., 2 * N + N * N + ia]
        return None
    aggs = []
    if op_code == OP_ADD1:
        aggs.append(_sa.AggCount_int64(g, 1, _T))
    else:
        w = wlist[0]
        pf = _postfix(w)
        if op_code == OP_MIN_MAX:
            kinds = [("AggMin_", None), ("AggMax_", None)]
        else:
            kinds = [("AggCount_", None), ("AggSum_", None), ("AggSumMoment_", 2)][:fields]
        for cls, moment in kinds:
            a = getattr(_sa, cls + pf)(g, 1, _T, moment) if moment is not None else getattr(_sa, cls + pf)(g, 1, _T)
            a.set_data(_SLOT, w, 0)
            aggs.append(a)
    if n:
        g.bin(_SLOT, aggs, n)
    for f, a in enumerate(aggs):
        part = np.asarray(a.get_result())[inner]
        if op_code == OP_MIN_MAX:
            np.minimum(grid[..., 0], part, out=grid[..., 0]) if f == 0 else np.maximum(grid[..., 1], part, out=grid[..., 1])
        else:
            grid[..., f] += part
    return None
