"""Host-side driver of the hot path: the part of vaex's DataFrame API that ends in `Grid.bin`.

`Frame` mirrors, for columns that are plain arrays, the call shapes of
    df.count/sum/mean/var/std/min/max/minmax(expression, binby=, limits=, shape=, selection=, edges=)
        (/root/reference/packages/vaex-core/vaex/dataframe.py:944-1245, _compute_agg :842-941)
    df.groupby(by, agg={...})                                   (vaex/groupby.py:602-1017)
and drives the `superagg` class surface exactly the way vaex's own task part does
(vaex/cpu.py:678-786): descriptors -> primitive aggregations (count / sum / summoment / min / max,
vaex/agg.py:386-523) -> ONE fused pass per call -> numpy finishers -> edge-cell slicing
(vaex/agg.py:323-335).  It exists so that the parity tests and bench read like vaex code on a box
where vaex itself is not installed; with vaex present, `vaex_amd.install()` makes vaex's unmodified
DataFrame do the same through the same classes.

Columns may be
  * numpy arrays (host): streamed in `chunk_size`-row chunks over `nthreads` slots — each slot stages its
    chunk to HBM on its own HIP stream, like the executor's thread pool (vaex/execution.py:432-435);
  * device arrays (anything with __cuda_array_interface__, e.g. torch cuda tensors): used in place.
Expressions are column names only: vaex's expression system is out of scope.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import predicate as _predicate
from . import superagg as _sa

_DTYPE_NAMES = {"float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"}


def _is_device(x):
    return hasattr(x, "__cuda_array_interface__") and not isinstance(x, np.ndarray)


def _class_postfix(col):
    """dtype name + '_non_native' the way vaex.utils.find_type_from_dtype builds it (vaex/utils.py:754-791)."""
    if _is_device(col):
        name = str(col.dtype).replace("torch.", "")
        return {"bool": "bool"}.get(name, name)
    dt = col.dtype
    if dt.kind in "mM":
        return "int64"
    name = dt.newbyteorder("=").name
    if name not in _DTYPE_NAMES:
        raise TypeError(f"unsupported dtype {dt}")
    return name + ("_non_native" if dt.byteorder == ">" else "")


def _as_u8(mask):
    if _is_device(mask):
        return mask
    m = np.ascontiguousarray(mask)
    return m.view(np.uint8) if m.dtype == np.bool_ else m.astype(np.uint8)



def _same_selection(a, b):
    """two selection arguments name the same rows: None / False are 'no selection', strings compare by value (an equal but distinct
    expression string must not drop a call to the slow path), anything else by identity"""
    a = None if a is False else a
    b = None if b is False else b
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, str) and isinstance(b, str):
        return a == b
    return a is b


class _Prim:
    """One native aggregation: (kind, column, selection, moment) — what AggregatorDescriptorBasic is to vaex."""

    def __init__(self, kind, column=None, selection=None, moment=0, as_float64=False):
        self.kind, self.column, self.selection, self.moment, self.as_float64 = kind, column, selection, moment, as_float64

    def key(self):
        return (self.kind, self.column, self.selection if isinstance(self.selection, (str, type(None))) else id(self.selection), self.moment, self.as_float64)


class agg:
    """Aggregation descriptors (the subset of vaex.agg on the hot path)."""

    class _Desc:
        def __init__(self, name, column=None, selection=None):
            self.name, self.column, self.selection = name, column, selection

        def prims(self):
            c, s = self.column, self.selection
            n = self.name
            if n == "count":
                return [_Prim("count", c, s)]
            if n == "sum":
                return [_Prim("sum", c, s)]
            if n == "min":
                return [_Prim("min", c, s)]
            if n == "max":
                return [_Prim("max", c, s)]
            if n == "mean":  # vaex/agg.py:391-418
                return [_Prim("sum", c, s), _Prim("count", c, s)]
            if n in ("var", "std"):  # vaex/agg.py:426-455: on astype(float64); ddof is ignored by the reference
                return [_Prim("summoment", c, s, 2, True), _Prim("sum", c, s, 0, True), _Prim("count", c, s, 0, True)]
            raise ValueError(n)

        def finish_spec(self, sa, aggs):
            """the same finish as a (op, agg0, agg1, agg2) tuple for the device finishers (superagg.finish)"""
            n = self.name
            if n in ("count", "sum", "min", "max"):
                return (sa.FIN_COPY, aggs[0], None, None)
            if n == "mean":
                return (sa.FIN_MEAN, aggs[0], aggs[1], None)
            return (sa.FIN_VAR if n == "var" else sa.FIN_STD, aggs[0], aggs[1], aggs[2])

        def finish(self, parts):
            n = self.name
            if n in ("count", "sum", "min", "max"):
                return parts[0]
            with np.errstate(divide="ignore", invalid="ignore"):
                if n == "mean":
                    return parts[0] / parts[1]
                mean = parts[1] / parts[2]
                variance = parts[0] / parts[2] - mean ** 2
                return variance if n == "var" else variance ** 0.5

    @staticmethod
    def count(column=None, selection=None):
        return agg._Desc("count", None if column in (None, "*") else column, selection)

    @staticmethod
    def sum(column, selection=None):
        return agg._Desc("sum", column, selection)

    @staticmethod
    def mean(column, selection=None):
        return agg._Desc("mean", column, selection)

    @staticmethod
    def var(column, selection=None):
        return agg._Desc("var", column, selection)

    @staticmethod
    def std(column, selection=None):
        return agg._Desc("std", column, selection)

    @staticmethod
    def min(column, selection=None):
        return agg._Desc("min", column, selection)

    @staticmethod
    def max(column, selection=None):
        return agg._Desc("max", column, selection)


_KIND_CLASS = {"count": "AggCount_", "sum": "AggSum_", "summoment": "AggSumMoment_", "min": "AggMin_", "max": "AggMax_"}


#: key dtypes the groupby's heavy-hitter peel handles (its row-wise intermediates are torch tensors: the dtypes torch computes with)
_PEEL_KEY_KINDS = ("int64", "int32", "int16", "int8", "uint8")


def _memo(col, value):
    """what a pass found out about a column, remembered WITH the column object (its address cannot be reused while it is held) and — torch
    tensors — its in-place version counter: vaex's columns are immutable by contract (vaex/dataframe.py caches rest on it), a caller's torch
    tensor is not (a tensor overwritten in place kept its old key range / NaN verdict)"""
    return (col, value, getattr(col, "_version", None))


def _memo_hit(hit, col):
    return hit is not None and hit[0] is col and hit[2] == getattr(col, "_version", None)


class Frame:
    def __init__(self, columns=None, chunk_size=1 << 20, nthreads=4, superagg=None, comm=None, **kw):
        """comm: a vaex_amd.dist.Comm when this Frame holds ONE RANK'S ROWS of a row-sharded table — every result is
        then the result over all ranks' rows (grids all-reduced before the finishers, min/max and key sets agreed)."""
        self.columns = dict(columns or {})
        self.columns.update(kw)
        self.comm = comm
        self.chunk_size = int(chunk_size)
        self.nthreads = int(nthreads)
        self.sa = superagg or _sa
        n = {len(c) for c in self.columns.values()}
        if len(n) > 1:
            raise ValueError("columns differ in length")
        self.n = n.pop() if n else 0
        self._f64_cache = {}
        self._predicates = {}
        self.direct_groupby_cells = 1 << 21  # widest key range binned without a hash map (fits the partition strategy)

    def __len__(self):
        return self.n

    # ------------------------------------------------------------------ column helpers
    def _col(self, name, as_float64=False):
        c = self.columns[name]
        if not as_float64:
            return c
        if not _memo_hit(self._f64_cache.get(name), c):
            if _is_device(c):
                conv = c.double()
            elif np.ma.isMaskedArray(c):
                conv = np.ma.array(np.ma.getdata(c).astype("f8"), mask=np.ma.getmaskarray(c))
            else:
                conv = c.astype("f8") if c.dtype != np.float64 else c
            self._f64_cache[name] = _memo(c, conv)
        return self._f64_cache[name][1]

    def _selection_mask(self, selection):
        """None, a mask array (host / device; 1 = keep), or a predicate.Predicate the GPU evaluates itself: a selection
        given as an expression string ("(x > 0) & (v < 3.5)"; a bare column name is that boolean column)"""
        if selection is None or selection is False:
            return None
        if isinstance(selection, str):
            if selection in self.columns:
                return self.columns[selection]
            if selection not in self._predicates:
                self._predicates[selection] = _predicate.compile_selection(selection, self.columns)
            return self._predicates[selection]
        return selection

    def _mask_array(self, selection):
        """the selection as a keep-mask array (the passes that have no device-side predicate: minmax, hashed groupby)"""
        sel = self._selection_mask(selection)
        if isinstance(sel, _predicate.Predicate):
            cols = {n: self.columns[n] for n in sel.columns}
            if any(_is_device(c) for c in cols.values()):
                mask = self._device_mask(sel, cols)
                return mask if mask is not None else sel.torch_mask(cols)   # (terms AND their arithmetic programs: ADVICE r5 — the programs used to be dropped here)
            return sel.numpy_mask(cols)
        return sel

    def _device_mask(self, pred, cols):
        """the predicate's keep bytes over device columns from ONE pass of the library's sel_eval (vxh_selection_evaluate; round 6) — the
        kernel the binned passes' device selections run through, so the rows kept are the same by construction — or None when the shim has no
        such entry or a column's dtype has no device comparison (then: Predicate.torch_mask)"""
        if not hasattr(getattr(self.sa, "Selection", None), "evaluate") or not all(_is_device(c) for c in cols.values()):
            return None
        try:
            import torch
            dtypes = [_predicate.dtype_code(cols[c].dtype) for c in pred.columns]
        except (ImportError, _predicate.Unsupported):
            return None
        first = cols[pred.columns[0]]
        n = len(first)
        cache = self.__dict__.setdefault("_device_masks", {})
        key = pred.key()
        hit = cache.get(key)
        if hit is not None and all(_memo_hit(m, cols[c]) for m, c in zip(hit[0], pred.columns)):
            return hit[1]
        sel = self.sa.Selection(1, dtypes, [(c, op, v) for c, op, v in pred.terms], pred.truth)
        if pred.programs:
            sel.set_programs({t: [tuple(st) for st in steps] for t, steps in pred.programs.items()})
        for i, c in enumerate(pred.columns):
            sel.set_data(0, i, cols[c])
        out = torch.empty((n + 3) & ~3, dtype=torch.uint8, device=first.device)
        sel.evaluate(0, n, out)
        self.sa.slot_wait(0)   # (the mask is also read by torch kernels on torch's stream — the peel's intermediates — which the slot's stream does not order)
        mask = out[:n]
        cache.clear()   # (one mask per frame at a time: n bytes of HBM each)
        cache[key] = ([_memo(cols[c], None) for c in pred.columns], mask)
        return mask

    # ------------------------------------------------------------------ binners
    def minmax(self, column, selection=None):
        """df.minmax: the legacy statisticNd OP_MIN_MAX pass (vaex/dataframe.py:1520, vaexfast.cpp:1090-1101)."""
        c = self.columns[column]
        sel = self._mask_array(selection)
        keep = None if sel is None else _as_u8(sel)
        if _is_device(c):
            pf = _class_postfix(c)
            if keep is not None and not _is_device(keep):  # the selection must live where the column lives
                import torch
                keep = torch.from_numpy(np.ascontiguousarray(keep)).cuda()
            elif keep is not None and str(keep.dtype).endswith("bool"):
                keep = keep.view(dtype=__import__("torch").uint8)
            return self._global_minmax(self.sa.minmax(c, keep, _DT_CODE[pf.replace("_non_native", "")], pf.endswith("_non_native")))
        data, miss = (np.ma.getdata(c), np.ma.getmaskarray(c)) if np.ma.isMaskedArray(c) else (c, None)
        if miss is not None:
            k = ~miss if keep is None else (keep.astype(bool) & ~miss)
            keep = _as_u8(k)
        pf = _class_postfix(data)
        return self._global_minmax(self.sa.minmax(np.ascontiguousarray(data), keep, _DT_CODE[pf.replace("_non_native", "")], pf.endswith("_non_native")))

    def _global_minmax(self, lohi):
        lo, hi = lohi
        if self.comm is not None:
            lo, hi = self.comm.minmax_float(float(lo), float(hi))
        return np.array([lo, hi])

    def _binner_specs(self, binby, limits, shape):
        binby = [binby] if isinstance(binby, (str, tuple, dict)) else list(binby or [])
        nd = len(binby)
        if not isinstance(shape, (list, tuple)):
            shape = [shape] * nd
        if limits is not None and nd == 1 and np.ndim(limits) == 1:
            limits = [limits]
        specs = []
        for d, b in enumerate(binby):
            if isinstance(b, dict):  # explicit ordinal binner: dict(column=, count=, min_value=0, invert=False)
                specs.append(dict(kind="ordinal", column=b["column"], count=int(b["count"]), min_value=int(b.get("min_value", 0)), invert=bool(b.get("invert", False))))
                continue
            lim = None if limits is None else limits[d]
            if lim is None or (isinstance(lim, str) and lim == "minmax"):
                lim = self.minmax(b)  # limits=None costs this extra pass in vaex too (dataframe.py:1927-1947)
            specs.append(dict(kind="scalar", column=b, vmin=float(lim[0]), vmax=float(lim[1]), bins=int(shape[d])))
        return specs

    # ------------------------------------------------------------------ the pass
    def _run_pass(self, specs, prims):
        """One fused pass: every primitive aggregation shares the bin index (TaskAggregations, vaex/tasks.py:473-580)."""
        sa = self.sa
        nthreads = max(1, self.nthreads)
        cols = {}

        def column(name, f64=False):
            key = (name, f64)
            if key not in cols:
                cols[key] = self._col(name, f64)
            return cols[key]

        bcols = [column(s["column"]) for s in specs]
        all_device = all(_is_device(c) for c in bcols) and all(p.column is None or _is_device(column(p.column, p.as_float64)) for p in prims)
        # selections given as expressions: one device-side Selection per distinct predicate (vaex_amd/predicate.py)
        sels = [self._selection_mask(p.selection) for p in prims]
        preds = {}
        for sel in sels:
            if isinstance(sel, _predicate.Predicate) and sel.key() not in preds:
                pcols = [column(n) for n in sel.columns]
                if any(np.ma.isMaskedArray(c) for c in pcols):
                    raise ValueError("selection over a column with missing values: pass the mask array instead")
                all_device = all_device and all(_is_device(c) for c in pcols)
                preds[sel.key()] = [None, pcols, [_predicate.dtype_code(c.dtype) for c in pcols], sel]
            elif sel is not None and not isinstance(sel, _predicate.Predicate):
                all_device = all_device and _is_device(sel)
        slots = 1 if all_device else nthreads
        binners = []
        for s, c in zip(specs, bcols):
            data = np.ma.getdata(c) if np.ma.isMaskedArray(c) else c
            pf = _class_postfix(data)
            if s["kind"] == "scalar":
                binners.append(getattr(sa, "BinnerScalar_" + pf)(slots, s["column"], s["vmin"], s["vmax"], s["bins"]))
            else:
                binners.append(getattr(sa, "BinnerOrdinal_" + pf)(slots, s["column"], s["count"], s["min_value"], False, s["invert"]))
        for entry in preds.values():
            entry[0] = sa.Selection(slots, entry[2], [(c, op, v) for c, op, v in entry[3].terms], entry[3].truth)
            if entry[3].programs:   # terms whose left side is an arithmetic expression over float64 columns (vxh_selection_set_program)
                entry[0].set_programs({t: [tuple(st) for st in steps] for t, steps in entry[3].programs.items()})
        grid = sa.Grid(binners)
        aggs = []
        for p in prims:
            if p.column is None:
                pf = "int64"  # count(*): vaex/agg.py:255-257
            else:
                c = column(p.column, p.as_float64)
                pf = _class_postfix(np.ma.getdata(c) if np.ma.isMaskedArray(c) else c)
            cls = getattr(sa, _KIND_CLASS[p.kind] + pf)
            aggs.append(cls(grid, slots, slots, p.moment) if p.kind == "summoment" else cls(grid, slots, slots))

        def process(slot, i1, i2):
            refs = []
            for b, c in zip(binners, bcols):
                if np.ma.isMaskedArray(c):
                    d, m = np.ascontiguousarray(np.ma.getdata(c)[i1:i2]), _as_u8(np.ma.getmaskarray(c)[i1:i2])
                    b.set_data(slot, d); b.set_data_mask(slot, m); refs += [d, m]
                else:
                    d = c[i1:i2] if _is_device(c) else np.ascontiguousarray(c[i1:i2])
                    b.set_data(slot, d); b.clear_data_mask(slot); refs.append(d)
            for entry in preds.values():
                for ci, c in enumerate(entry[1]):
                    d = c[i1:i2] if _is_device(c) else np.ascontiguousarray(c[i1:i2])
                    entry[0].set_data(slot, ci, d.view("u1") if (not _is_device(d) and d.dtype == np.bool_) else d); refs.append(d)
            for a, p, sel in zip(aggs, prims, sels):
                if isinstance(sel, _predicate.Predicate):
                    a.set_selection(preds[sel.key()][0])
                    sel = None
                mask = None if sel is None else sel[i1:i2]
                if p.column is not None:
                    c = column(p.column, p.as_float64)
                    if np.ma.isMaskedArray(c):  # missing values: selection & ~mask (vaex/cpu.py:770-784)
                        d, miss = np.ascontiguousarray(np.ma.getdata(c)[i1:i2]), np.ma.getmaskarray(c)[i1:i2]
                        mask = ~miss if mask is None else (np.asarray(mask).astype(bool) & ~miss)
                    else:
                        d = c[i1:i2] if _is_device(c) else np.ascontiguousarray(c[i1:i2])
                    a.set_data(slot, d, 0); refs.append(d)
                if mask is not None:
                    m = _as_u8(mask)
                    a.set_data_mask(slot, m); refs.append(m)
                else:
                    a.clear_data_mask(slot)
            grid.bin(slot, aggs, i2 - i1)
            return refs

        n = self.n
        self.last_slots = [0]   # the thread slots the most recent pass binned on (sa.last_kernel(slot) says what ran there)
        if n:
            if all_device:
                process(0, 0, n)
            else:
                free = list(range(slots))
                lock = threading.Lock()
                used = set()

                def work(i1):
                    with lock:
                        slot = free.pop()
                        used.add(slot)
                    try:
                        process(slot, i1, min(n, i1 + self.chunk_size))
                    finally:
                        with lock:
                            free.append(slot)
                with ThreadPoolExecutor(slots) as pool:
                    list(pool.map(work, range(0, n, self.chunk_size)))
                self.last_slots = sorted(used) or [0]
        return grid, aggs

    def _pass(self, descs, binby=None, limits=None, shape=128, reduce=None):
        """descriptors -> distinct primitive aggregations -> ONE fused pass (+ the cross-rank reduce); returns
        (binner specs, grid, aggregator objects, per descriptor the indices of its primitives)"""
        specs = self._binner_specs(binby, limits, shape)
        prims, index = [], {}
        want = []
        for d in descs:
            ids = []
            for p in d.prims():
                if p.as_float64 and p.column is not None and str(self.columns[p.column].dtype).replace("torch.", "") == "float64":
                    p.as_float64 = False  # astype('float64') of a float64 column is the column: std's sum / count are mean's
                k = p.key()
                if k not in index:
                    index[k] = len(prims)
                    prims.append(p)
                ids.append(index[k])
            want.append(ids)
        grid, aggs = self._run_pass(specs, prims)
        if reduce is None and self.comm is not None:
            reduce = self.comm.allreduce
        if reduce is not None:
            from .dist import with_reduced
            aggs = with_reduced(aggs, reduce(aggs))   # (aggregators that cannot be written through come back as result arrays)
        return specs, grid, aggs, want

    def _agg(self, descs, binby=None, limits=None, shape=128, edges=False, reduce=None):
        """Evaluate a list of descriptors in ONE pass; returns one ndarray per descriptor.  reduce: optional
        callable(list of aggregator objects) run before the results are read (multi-GPU all-reduce hook)."""
        specs, grid, aggs, want = self._pass(descs, binby, limits, shape, reduce)
        raw = [np.asarray(a.get_result()) for a in aggs]  # (get_result already returns a fresh array)
        if not edges:  # vaex/agg.py:323-335
            sl = tuple(slice(2, -1) if s["kind"] == "scalar" else slice(0, -2) for s in specs)
            raw = [r[sl] for r in raw]
        return [d.finish([raw[i] for i in ids]) for d, ids in zip(descs, want)]

    # ------------------------------------------------------------------ the DataFrame-like methods
    def _one(self, name, expression, binby, limits, shape, selection, edges):
        desc = getattr(agg, name)(expression, selection=selection) if name != "count" else agg.count(expression, selection=selection)
        return self._agg([desc], binby, limits, shape, edges)[0]

    def count(self, expression=None, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._one("count", expression, binby, limits, shape, selection, edges)

    def sum(self, expression, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._one("sum", expression, binby, limits, shape, selection, edges)

    def mean(self, expression, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._one("mean", expression, binby, limits, shape, selection, edges)

    def var(self, expression, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._one("var", expression, binby, limits, shape, selection, edges)

    def std(self, expression, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._one("std", expression, binby, limits, shape, selection, edges)

    def min(self, expression, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._one("min", expression, binby, limits, shape, selection, edges)

    def max(self, expression, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._one("max", expression, binby, limits, shape, selection, edges)

    def _result_dtype(self, desc):
        """dtype of a descriptor's result column as the reference returns it (vaex/agg.py: count int64, sum upcast,
        min/max the input's type, mean/var/std float64)"""
        n = desc.name
        if n == "count":
            return np.dtype("int64")
        if n in ("mean", "var", "std"):
            return np.dtype("float64")
        c = self.columns[desc.column]
        name = str(c.dtype).replace("torch.", "")
        dt = np.dtype("bool" if name == "bool" else name)
        if n == "sum":
            return np.dtype("float64") if dt.kind == "f" else (np.dtype("uint64") if dt.kind == "u" else np.dtype("int64"))
        return dt.newbyteorder("=")

    # ------------------------------------------------------------------ groupby
    # ------------------------------------------------------------------ limits from the data
    def limits_percentage(self, column, percentage=99.73, selection=None):
        """df.limits_percentage (vaex/dataframe.py:1795-1840): the [lo, hi] range around the median that holds
        `percentage` % of the rows — a min/max pass, a 16384-bin 1-d count pass (both on the GPU), then the
        reference's numpy: cumulative counts, linear interpolation."""
        vmin, vmax = self.minmax(column, selection=selection)
        size = 1024 * 16
        counts = self.count(binby=column, shape=size, limits=[vmin, vmax], selection=selection)
        cumcounts = np.concatenate([[0], np.cumsum(counts)])
        cumcounts = cumcounts / cumcounts.max()
        f = (1 - percentage / 100.) / 2
        x = np.linspace(vmin, vmax, size + 1)
        return np.interp([f, 1 - f], cumcounts, x)

    def percentile_approx(self, column, percentage=50., binby=(), limits=None, shape=128, percentile_shape=1024, percentile_limits="minmax", selection=None):
        """df.percentile_approx (vaex/dataframe.py:1632-1760): the cumulative distribution of `column` on a grid of
        percentile_shape cells (per binby cell), searched for the percentile and interpolated.  The count pass
        (binby + [column], edges=True) runs on the GPU; the finish restates the reference's numpy and its
        `vaexfast.grid_find_edges` (src/vaexfast.cpp:1680-1719) on the small result grid."""
        binby = [binby] if isinstance(binby, str) else list(binby)
        if isinstance(percentile_limits, str):
            if percentile_limits != "minmax":
                raise NotImplementedError("percentile_limits must be 'minmax' or [lo, hi]")
            plim = self.minmax(column, selection=selection)
            if str(self.columns[column].dtype).replace("torch.", "") == "float32":
                # (vaex's minmax comes back in the column's own type — vaex/dataframe.py:1528 — so for a float32 column `ub - lb` below is a
                #  float32 subtraction there: mirrored, or the interpolated percentile differs in the 7th digit)
                plim = np.asarray(plim).astype(np.float32)
        else:
            plim = percentile_limits
        nb = len(binby)
        if not isinstance(shape, (list, tuple)):
            shape = [shape] * nb
        if nb:
            if limits is not None and nb == 1 and np.ndim(limits) == 1:
                limits = [limits]
            blim = [self.minmax(b) if (limits is None or limits[i] is None) else limits[i] for i, b in enumerate(binby)]
        else:
            blim = []
        counts = self.count(binby=binby + [column], shape=list(shape) + [percentile_shape], limits=list(blim) + [list(plim)], selection=selection, edges=True)
        nonnans = tuple([slice(2, -1)] * nb + [slice(1, None)])  # drop the edge cells of the binby dims and the NaN cell of the last
        counts_nonnans = counts[nonnans]
        cumulative_grid = np.cumsum(counts_nonnans, -1).astype(np.float64)
        totalcounts = np.sum(counts_nonnans, -1)
        empty = totalcounts == 0
        size = cumulative_grid.shape[-1]

        def find_edges(values):  # per grid row: last edge left of `value`, first edge not left of it
            edges = np.zeros(cumulative_grid.shape[:-1] + (2,), dtype=np.int64)
            for i in np.ndindex(cumulative_grid.shape[:-1]):
                row, value = cumulative_grid[i], values[i]
                left = 0
                while left < size - 1 and row[left + 1] < value:
                    left += 1
                right = left
                while right < size - 1 and row[right] < value:
                    right += 1
                edges[i] = (left, right)
            return edges

        def index_choose(a, indices):
            out = np.zeros(a.shape[:-1])
            for i in np.ndindex(out.shape):
                out[i] = a[i + (indices[i],)]
            return out

        lb, ub = plim
        waslist = isinstance(percentage, (list, tuple, np.ndarray))
        percentiles = []
        for p in (percentage if waslist else [percentage]):
            if p == 0:
                percentiles.append(lb)
                continue
            if p == 100:
                percentiles.append(ub)
                continue
            values = np.array((totalcounts + 1) * p / 100.)
            values[empty] = 0
            floor_values, ceil_values = np.array(np.floor(values)), np.array(np.ceil(values))

            def calculate_x(edges, vals):
                left, right = edges[..., 0], edges[..., 1]
                left_value, right_value = index_choose(cumulative_grid, left), index_choose(cumulative_grid, right)
                denom = np.array(right_value - left_value)
                denom[denom == 0] = 1.0
                u = np.array(vals - left_value) / denom
                xleft = lb + (left - 0.5) * (ub - lb) / (size - 3)
                xright = lb + (right - 0.5) * (ub - lb) / (size - 3)
                return xleft + (xright - xleft) * u

            x1 = calculate_x(find_edges(floor_values), floor_values)
            x2 = calculate_x(find_edges(ceil_values), ceil_values)
            percentiles.append(x1 + (x2 - x1) * (values - floor_values))
        return np.array(percentiles) if waslist else np.array(percentiles[0])

    def median_approx(self, column, binby=(), limits=None, shape=128, percentile_shape=256, percentile_limits="minmax", selection=None):
        return self.percentile_approx(column, 50, binby=binby, limits=limits, shape=shape, percentile_shape=percentile_shape, percentile_limits=percentile_limits, selection=selection)

    # ------------------------------------------------------------------ first / last
    def _first_last(self, invert, expression, order_expression, binby, limits, shape, selection, edges):
        """df.first / df.last (vaex/dataframe.py:976-1011 -> AggFirst, vaex/agg.py:556-576): per cell the value of the row
        with the smallest (last: largest) order value; a masked array (cells without rows masked)."""
        comm = self.comm
        sa = self.sa
        specs = self._binner_specs(binby, limits, shape)
        value = self.columns[expression]
        order = None if order_expression is None else self.columns[order_expression]
        cols = [self.columns[s["column"]] for s in specs] + [value] + ([order] if order is not None else [])
        if any(np.ma.isMaskedArray(c) for c in cols):
            raise NotImplementedError("first/last over columns with missing values")
        sel = self._mask_array(selection)
        device = all(_is_device(c) for c in cols) and (sel is None or _is_device(sel))
        binners = []
        for s, c in zip(specs, cols):
            pf = _class_postfix(c)
            if s["kind"] == "scalar":
                binners.append(getattr(sa, "BinnerScalar_" + pf)(1, s["column"], s["vmin"], s["vmax"], s["bins"]))
            else:
                binners.append(getattr(sa, "BinnerOrdinal_" + pf)(1, s["column"], s["count"], s["min_value"], False, s["invert"]))
        grid = sa.Grid(binners)
        opf = "int64" if order is None else _class_postfix(order)  # vaex/agg.py:280-283: "rows use int64"
        nn = "_non_native" if _class_postfix(value).endswith("_non_native") else ""
        a = getattr(sa, "AggFirst_" + _class_postfix(value).replace("_non_native", "") + "_" + opf.replace("_non_native", "") + nn)(grid, 1, 1, invert)
        # a row-sharded frame also needs every cell's winning ORDER value: the product hands it out (raw_result); classes without
        # that accessor (the reference's) get a second AggFirst whose value column is the order column itself
        a2 = None
        if comm is not None and not hasattr(a, "raw_result"):
            if order is None:
                raise NotImplementedError("first / last without an order column over a row-sharded Frame needs the aggregator's raw_result()")
            a2 = getattr(sa, "AggFirst_" + opf.replace("_non_native", "") + "_" + opf.replace("_non_native", "") + nn)(grid, 1, 1, invert)
        step = self.n if device else self.chunk_size
        if order is None and step < self.n:
            raise NotImplementedError("first/last without an order column over several chunks: the order would be the row's index inside its chunk (src/agg_first.cpp:136)")
        for i1 in range(0, self.n, max(1, step)):
            i2 = min(self.n, i1 + step)
            refs = []
            pick = (lambda c: c[i1:i2]) if device else (lambda c: (lambda d: d.view("u1") if d.dtype == np.bool_ else d)(np.ascontiguousarray(c[i1:i2])))
            for b, c in zip(binners, cols):
                d = pick(c); b.set_data(0, d); b.clear_data_mask(0); refs.append(d)
            d = pick(value); a.set_data(0, d, 0); refs.append(d)
            if order is not None:
                d = pick(order); a.set_data(0, d, 1); refs.append(d)
            if a2 is not None:
                d = pick(order); a2.set_data(0, d, 0); a2.set_data(0, d, 1); refs.append(d)
            if sel is not None:
                m = _as_u8(sel[i1:i2]); refs.append(m)
                for x in (a, a2):
                    if x is not None:
                        x.set_data_mask(0, m)
            else:
                for x in (a, a2):
                    if x is not None:
                        x.clear_data_mask(0)
            grid.bin(0, [a] if a2 is None else [a, a2], i2 - i1)
        r = a.get_result() if comm is None else self._first_last_allranks(a, a2, order is None, invert, comm)
        if not edges:
            r = r[tuple(slice(2, -1) if s["kind"] == "scalar" else slice(0, -2) for s in specs)]
        return r

    def _first_last_allranks(self, a, a2, by_row_index, invert, comm):
        """per cell the winner over ALL ranks' rows (the cross-rank form of the merge the reference's AggFirst does not have,
        src/agg_first.cpp:42): MIN (last: MAX) all-reduce of the ranks' winning order values, MIN all-reduce of the rank that holds
        it (equal orders: the earlier rows win, as `value_order < grid_data_order[i]` keeps the first one seen), and the
        winner's value as the only non-zero term of a SUM all-reduce of bit patterns.  Three grid-sized all-reduces."""
        if a2 is None:
            values, masked, order = (np.array(x) for x in a.raw_result())
        else:
            r1, r2 = a.get_result(), a2.get_result()
            values, masked, order = np.array(np.ma.getdata(r1)), np.array(np.ma.getmaskarray(r1)), np.array(np.ma.getdata(r2))
        masked = masked.astype(bool)
        rank, world = comm.rank(), comm.world()
        if order.dtype == np.uint64:
            raise NotImplementedError("first / last ordered by a uint64 column over a row-sharded Frame")
        if by_row_index:  # the order is the row's index: make it global (rows are sharded in rank order)
            counts = [int(p[0][0]) for p in comm.all_gather_arrays([np.array([self.n], dtype=np.int64)])]
            order = order.astype(np.int64) + sum(counts[:rank])
        is_float = order.dtype.kind == "f"
        o = order.astype(np.float64 if is_float else np.int64)
        worst = (np.inf if is_float else np.iinfo(np.int64).max) if not invert else (-np.inf if is_float else np.iinfo(np.int64).min)
        o = np.where(masked, worst, o)
        best = comm.allreduce_arrays([o], ["max" if invert else "min"])[0]
        cand = np.where(~masked & (o == best), rank, world).astype(np.int64)
        winner = comm.allreduce_arrays([cand], ["min"])[0]
        mine = winner == rank
        v = np.ascontiguousarray(values)
        if v.dtype.itemsize == 8:
            bits = v.view(np.int64)
        elif v.dtype.kind == "f":
            bits = v.view(np.int32).astype(np.int64)
        else:
            bits = v.astype(np.int64)
        bits = comm.allreduce_arrays([np.where(mine, bits, 0)], ["sum"])[0]
        if v.dtype.itemsize == 8:
            out = bits.view(v.dtype)
        elif v.dtype.kind == "f":
            out = bits.astype(np.int32).view(np.float32)
        else:
            out = bits.astype(v.dtype)
        empty = winner == world
        out = np.array(out)
        out[empty] = True if out.dtype == np.bool_ else 99  # (what the reference's empty cells read under the mask, src/agg_first.cpp:22-28)
        return np.ma.array(out, mask=empty)

    def first(self, expression, order_expression=None, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._first_last(False, expression, order_expression, binby, limits, shape, selection, edges)

    def last(self, expression, order_expression=None, binby=None, limits=None, shape=128, selection=None, edges=False):
        return self._first_last(True, expression, order_expression, binby, limits, shape, selection, edges)

    def nunique(self, expression, binby=None, limits=None, shape=128, selection=None, dropna=False, dropnan=False, dropmissing=False, edges=False):
        """df.nunique (vaex/dataframe.py:1057-1089 -> vaex.agg.nunique, vaex/agg.py:600-612 -> AggNUnique_<T>): per cell the
        number of distinct values of `expression`; NaN and the missing value count as one value each unless dropped
        (dropna = both).  The rows' {value, cell} pairs are sorted and reduced on the device (vxh_collect_*)."""
        sa = self.sa
        specs = self._binner_specs(binby or [], limits, shape)
        value = self.columns[expression]
        missing = None
        if np.ma.isMaskedArray(value):
            missing = np.ma.getmaskarray(value)
            value = np.ma.getdata(value)
        cols = [self.columns[s["column"]] for s in specs]
        if any(np.ma.isMaskedArray(c) for c in cols):
            raise NotImplementedError("nunique binned by columns with missing values")
        sel = self._mask_array(selection)
        device = all(_is_device(c) for c in cols + [value]) and (sel is None or _is_device(sel))
        binners = []
        for s_, c in zip(specs, cols):
            pf = _class_postfix(c)
            if s_["kind"] == "scalar":
                binners.append(getattr(sa, "BinnerScalar_" + pf)(1, s_["column"], s_["vmin"], s_["vmax"], s_["bins"]))
            else:
                binners.append(getattr(sa, "BinnerOrdinal_" + pf)(1, s_["column"], s_["count"], s_["min_value"], False, s_["invert"]))
        grid = sa.Grid(binners)
        make = lambda: getattr(sa, "AggNUnique_" + _class_postfix(value))(grid, 1, 1, bool(dropmissing or dropna), bool(dropnan or dropna))
        a = make()
        step = self.n if device else self.chunk_size
        for i1 in range(0, self.n, max(1, step)):
            i2 = min(self.n, i1 + step)
            refs = []
            pick = (lambda c: c[i1:i2]) if device else (lambda c: (lambda d: d.view("u1") if d.dtype == np.bool_ else d)(np.ascontiguousarray(c[i1:i2])))
            for b, c in zip(binners, cols):
                d = pick(c); b.set_data(0, d); b.clear_data_mask(0); refs.append(d)
            d = pick(value); a.set_data(0, d, 0); refs.append(d)
            # vaex/cpu.py:733-770: the selection goes in as selection mask, selection & ~missing as data mask
            if sel is not None:
                m = _as_u8(sel[i1:i2]); a.set_selection_mask(0, m); refs.append(m)
            else:
                a.clear_selection_mask(0)
            if missing is not None:
                m = _as_u8(~missing[i1:i2] if sel is None else (np.asarray(sel[i1:i2]).astype(bool) & ~missing[i1:i2])); a.set_data_mask(0, m); refs.append(m)
            elif sel is not None:
                a.set_data_mask(0, refs[-1])
            else:
                a.clear_data_mask(0)
            grid.bin(0, [a], i2 - i1)
        if self.comm is not None:
            a = self._collect_allranks(a, make, self.comm)
        r = np.asarray(a.get_result())
        if not specs:
            return int(r.reshape(-1)[0])
        if not edges:
            r = r[tuple(slice(2, -1) if s_["kind"] == "scalar" else slice(0, -2) for s_ in specs)]
        return r

    def list(self, expression, binby=None, limits=None, shape=128, selection=None, dropna=False, dropnan=False, dropmissing=False, edges=False):
        """vaex.agg.list (vaex/agg.py:655-670 -> AggList_<T>_<T2>): per cell the values of `expression` in row order, then one NaN
        per NaN row, then one (zero) slot per missing value (unless dropped).  Returns an object array of the grid's shape whose
        elements are the cells' value arrays (vaex wraps the same (offsets, values) pair into an arrow list array)."""
        sa = self.sa
        specs = self._binner_specs(binby or [], limits, shape)
        value = self.columns[expression]
        missing = None
        if np.ma.isMaskedArray(value):
            missing = np.ma.getmaskarray(value)
            value = np.ma.getdata(value)
        cols = [self.columns[s_["column"]] for s_ in specs]
        if any(np.ma.isMaskedArray(c) for c in cols):
            raise NotImplementedError("list binned by columns with missing values")
        sel = self._mask_array(selection)
        device = all(_is_device(c) for c in cols + [value]) and (sel is None or _is_device(sel))
        binners = []
        for s_, c in zip(specs, cols):
            pf = _class_postfix(c)
            if s_["kind"] == "scalar":
                binners.append(getattr(sa, "BinnerScalar_" + pf)(1, s_["column"], s_["vmin"], s_["vmax"], s_["bins"]))
            else:
                binners.append(getattr(sa, "BinnerOrdinal_" + pf)(1, s_["column"], s_["count"], s_["min_value"], False, s_["invert"]))
        grid = sa.Grid(binners)
        pf = _class_postfix(value)
        make = lambda: getattr(sa, "AggList_" + pf.replace("_non_native", "") + "_int64" + ("_non_native" if pf.endswith("_non_native") else ""))(grid, 1, 1, bool(dropnan or dropna), bool(dropmissing or dropna))
        a = make()
        step = self.n if device else self.chunk_size
        for i1 in range(0, self.n, max(1, step)):
            i2 = min(self.n, i1 + step)
            refs = []
            pick = (lambda c: c[i1:i2]) if device else (lambda c: (lambda d: d.view("u1") if d.dtype == np.bool_ else d)(np.ascontiguousarray(c[i1:i2])))
            for b, c in zip(binners, cols):
                d = pick(c); b.set_data(0, d); b.clear_data_mask(0); refs.append(d)
            d = pick(value); a.set_data(0, d, 0); refs.append(d)
            # data mask: 1 = value present (and selected), 0 = missing; rows outside the selection carry 2: neither kept nor counted
            # (vaex hands selection & ~missing as data mask, which turns unselected rows into missing ones: vaex/cpu.py:733-770)
            if sel is not None or missing is not None:
                m = np.ones(i2 - i1, dtype="u1")
                if missing is not None:
                    m[missing[i1:i2]] = 0
                if sel is not None:
                    m[~np.asarray(sel[i1:i2].cpu() if _is_device(sel) else sel[i1:i2]).astype(bool)] = 2
                if device:
                    import torch
                    m = torch.from_numpy(m).to(value.device)
                a.set_data_mask(0, m); refs.append(m)
            else:
                a.clear_data_mask(0)
            grid.bin(0, [a], i2 - i1)
        if self.comm is not None:
            a = self._collect_allranks(a, make, self.comm)
        offsets, values = a.list_arrays()
        offsets, values = np.asarray(offsets), np.asarray(values)
        cells = np.empty(len(offsets) - 1, dtype=object)
        for j in range(len(cells)):
            cells[j] = values[offsets[j]:offsets[j + 1]]
        if not specs:
            return cells[0]
        r = cells.reshape([len(b) for b in binners][::-1]).T  # (dim 0 fastest)
        if not edges:
            r = r[tuple(slice(2, -1) if s_["kind"] == "scalar" else slice(0, -2) for s_ in specs)]
        return r

    def _collect_allranks(self, a, make, comm):
        """AggNUnique / AggList over ALL ranks' rows: every rank's collector state — its {value bits, cell} pairs (nunique: the
        distinct ones) and the per-cell missing / NaN row counts (vxh_collect_pairs) — is all-gathered as device tensors and
        merged, in rank order (= row order: what AggList's per-cell sequences need), into a fresh collector on every rank
        (vxh_collect_merge_pairs).  The cross-rank form of TaskPartAggregation.reduce (vaex/cpu.py:788-796); the reference's own
        merge of these aggregators is empty / throws."""
        if not hasattr(a, "pairs"):
            raise NotImplementedError("nunique / list over a row-sharded Frame need the collector's pairs() / merge_pairs()")
        parts = comm.all_gather_arrays([np.asarray(x) for x in a.pairs()])
        merged = make()
        for p in parts:
            merged.merge_pairs(*p)
        return merged

    def value_counts(self, expression, dropna=False, dropnan=False, dropmissing=False, ascending=False):
        """df[expression].value_counts() (vaex/expression.py:1029-1130 -> TaskPartValueCounts, vaex/cpu.py:141-283: a
        `counter_<T>` hash map per thread, merged) for an integer or float column: (values, counts) sorted by count
        (descending unless `ascending`; ties by value).  = the groupby count of the column on itself: the dense-range
        ordinal pass or the fused hash aggregation on the device; float values are counted by their bits (-0.0 and +0.0 apart,
        NaN one value)."""
        col = self.columns[expression]
        missing = 0
        if np.ma.isMaskedArray(col):
            m = np.ma.getmaskarray(col)
            missing = int(m.sum())
            col = np.ma.getdata(col)[~m]
        kind = str(col.dtype).replace("torch.", "")
        nans = 0
        if kind.startswith("float"):
            import torch
            dev = _is_device(col)
            t = col if dev else torch.from_numpy(np.ascontiguousarray(col))
            isnan = torch.isnan(t)
            nans = int(isnan.sum())
            t = t[~isnan].to(torch.float64)
            keys = t.view(torch.int64)  # (by bit pattern: -0.0 and +0.0 are two values, as for the reference's counter)
            keys = keys if dev else keys.numpy()
        else:
            keys = col
        # (a row-sharded frame: the groupby on the sub-frame agrees on the keys and all-reduces the counts; the NaN / missing
        #  row counts are summed over the ranks)
        sub = Frame({"k": keys}, chunk_size=self.chunk_size, nthreads=self.nthreads, superagg=self.sa, comm=self.comm)
        if self.comm is not None:
            nans, missing = self.comm.sum_ints([nans, missing])
        out = sub.groupby("k", {"n": agg.count()}) if (len(keys) or self.comm is not None) else {"k": np.array([], dtype="i8"), "n": np.array([], dtype="i8")}
        values, counts = np.asarray(out["k"]), np.asarray(out["n"]).astype(np.int64)
        if kind.startswith("float"):
            values = values.astype(np.int64).view(np.float64).astype(kind)
            if nans and not (dropnan or dropna):
                values = np.concatenate([values, [np.nan]]); counts = np.concatenate([counts, [nans]])
        values = np.ma.array(values, mask=np.zeros(len(values), bool)) if missing and not (dropmissing or dropna) else values
        if missing and not (dropmissing or dropna):
            values = np.ma.concatenate([values, np.ma.masked_all(1, dtype=values.dtype)]); counts = np.concatenate([counts, [missing]])
        order = np.lexsort((np.arange(len(counts)), counts if ascending else -counts))
        return values[order], counts[order]

    # ------------------------------------------------------------------ groupby
    def _groupby_combined(self, by, agg_spec, reduce, comm, selection=None):
        """df.groupby([k1, k2, ...]): the keys' ordinals packed into ONE int64 key on the device (vxh_pack_keys — the
        expression vaex's GrouperCombined evaluates with numpy, vaex/groupby.py:526-584), the single-key machinery on the
        packed column, the packed group keys unpacked again.  Groups come out in ascending (k1, k2, ...) order."""
        import torch
        sa = self.sa
        cols, dtypes, mins, counts = [], [], [], []
        for k in by:
            c = self.columns[k]
            if np.ma.isMaskedArray(c):
                raise NotImplementedError("masked group keys")
            pf = _class_postfix(c)
            if pf.startswith("float") or pf.endswith("_non_native"):
                raise NotImplementedError("groupby on float / non-native keys")
            data = c if _is_device(c) else np.ascontiguousarray(c.view("u1") if c.dtype == np.bool_ else c)
            kmin, kmax = sa.minmax_int(data, None, _DT_CODE[pf], False) if self.n else ((0, 0) if comm is None else (2**63 - 1, -2**63))
            if comm is not None:   # ranks pack with the SAME minima and multipliers: the global key ranges
                kmin, kmax = comm.minmax(int(kmin), int(kmax))
                if kmin > kmax:
                    kmin, kmax = 0, 0
            cols.append(data); dtypes.append(_DT_CODE[pf]); mins.append(int(kmin)); counts.append(int(kmax) - int(kmin) + 1)
        total = 1
        for n in counts:
            total *= n
        if total >= 2**63 - 1:  # vaex/groupby.py:541-549 combines in stages through a hash map there; not built here
            raise NotImplementedError("the key ranges' product overflows 64 bits")
        mults = [1] * len(by)
        for i in range(len(by) - 2, -1, -1):
            mults[i] = mults[i + 1] * counts[i + 1]
        packed = sa.pack_keys(cols, dtypes, mins, mults)
        sub = {"__packed__": torch.as_tensor(packed, device="cuda")}
        for d in agg_spec.values():
            if d.column is not None and d.column not in sub:
                c = self.columns[d.column]
                if np.ma.isMaskedArray(c):
                    raise NotImplementedError("multi-key groupby over columns with missing values")
                sub[d.column] = c if _is_device(c) else torch.from_numpy(np.ascontiguousarray(c)).cuda()
            if d.selection is not None:
                raise NotImplementedError("multi-key groupby with a selection")
        if selection is not None:   # a filter over the call: its predicate's columns (or its mask) go along
            sel = self._selection_mask(selection)
            if isinstance(sel, _predicate.Predicate):
                for c in sel.columns:
                    if c not in sub:
                        col = self.columns[c]
                        if np.ma.isMaskedArray(col):
                            raise NotImplementedError("filter over a column with missing values")
                        sub[c] = col if _is_device(col) else torch.from_numpy(np.ascontiguousarray(col)).cuda()
            else:
                sub["__keep__"] = sel if _is_device(sel) else torch.from_numpy(np.ascontiguousarray(_as_u8(sel))).cuda()
                selection = "__keep__"
        f = Frame(sub, chunk_size=self.chunk_size, nthreads=self.nthreads, superagg=sa, comm=comm)
        f.direct_groupby_cells = self.direct_groupby_cells
        res = f.groupby("__packed__", agg_spec, reduce=reduce, comm=comm, selection=selection)
        self.last_groupby_info = getattr(f, "last_groupby_info", None)
        pk = np.asarray(res.pop("__packed__")).astype(np.int64)
        out = {}
        for k, mult, n, kmin in zip(by, mults, counts, mins):
            out[k] = ((pk // mult) % n + kmin).astype(np.asarray(self.columns[k][:0].cpu() if _is_device(self.columns[k]) else self.columns[k][:0]).dtype)
        out.update(res)
        del packed
        return out

    def _key_range(self, by, key, pf):
        """exact integer (min, max) of this rank's rows of a key column: one streaming pass (vxh_minmax_int), remembered per
        column — a frame's columns do not change under it, which is the contract vaex's own per-dataframe caches rest on
        (vaex/dataframe.py:1795-1840 `limits`, vaex/cache.py)."""
        if not self.n:
            return 2**63 - 1, -2**63
        cache = self.__dict__.setdefault("_key_range_cache", {})
        hit = cache.get(by)
        if _memo_hit(hit, key):
            return hit[1]
        r = self.sa.minmax_int(key if _is_device(key) else np.ascontiguousarray(key), None, _DT_CODE[pf], False)
        cache[by] = _memo(key, r)
        return r

    def _prescan(self, by, key, pf, descs):
        """A FIRST groupby over fresh device columns needs the key's exact range and whether a float64 value column holds NaN; both are
        remembered per column object, and both used to be a pass of their own (vxh_minmax_int 8 B/row; torch.isnan + any: 8 B/row read,
        1 B/row written and read back).  An int64 key and a float64 value column that are both unknown are scanned in ONE pass over
        the 16 bytes of a row (vxh_scan_key_value), which fills both memories."""
        if pf != "int64" or not self.n or not _is_device(key) or not hasattr(self.sa, "scan_key_value"):
            return
        kr = self.__dict__.setdefault("_key_range_cache", {})
        hit = kr.get(by)
        if _memo_hit(hit, key):
            return
        nc = self.__dict__.setdefault("_nan_cache", {})
        for d in descs:
            c = d.column
            if c is None:
                continue
            col = self.columns[c]
            if np.ma.isMaskedArray(col) or not _is_device(col) or str(col.dtype).replace("torch.", "") != "float64" or len(col) != len(key):
                continue
            seen = nc.get(c)
            if _memo_hit(seen, col):
                continue
            try:
                kmin, kmax, nans = self.sa.scan_key_value(key, col)
            except RuntimeError:   # (unaligned views: the two separate passes)
                return
            kr[by] = _memo(key, (int(kmin), int(kmax)))
            nc[c] = _memo(col, bool(nans))
            return

    def _may_hold_nan(self, name):
        """False when column `name` certainly holds no NaN / missing value: integer columns, and float columns scanned once
        (remembered per column object, like the key range)."""
        col = self.columns[name]
        if np.ma.isMaskedArray(col):
            return True
        kind = str(col.dtype).replace("torch.", "")
        if not kind.startswith("float"):
            return False
        cache = self.__dict__.setdefault("_nan_cache", {})
        hit = cache.get(name)
        if _memo_hit(hit, col):
            return hit[1]
        if _is_device(col):
            import torch
            r = bool(torch.isnan(col).any())
        else:
            # a host column: a numpy scan of it costs more than the extra aggregator it could save (1e8 rows: ~60 ms on one core
            # against a pass that is PCIe-bound either way) — assume it may
            r = True
        cache[name] = _memo(col, r)
        return r

    def groupby(self, by, agg_spec, reduce=None, comm=None, selection=None):
        """df.groupby(by).agg({...}) for ONE integer key column (a list of key columns: see _groupby_combined).
        Returns {by: keys (ascending), name: values}.  selection: a keep-filter over the whole call (an expression of
        vaex_amd.predicate's subset or a mask array) — what a filtered vaex frame is, `df[df.x > 0].groupby(...)`: rows outside it
        reach no aggregator and groups without a row inside it do not exist.

        The reference first collects the distinct keys (ordered_set, vaex/hash.py:152-171) and, when they are
        dense (range <= 4/3 * n_unique), simplifies to BinnerInteger (vaex/groupby.py:263-272).  Here the range
        is measured first (one streaming min/max pass): a range of up to `direct_groupby_cells` cells bins
        directly through BinnerOrdinal(min_value) in a single pass; wider ranges go through the GPU hash map
        inside the binner (BinnerHash).  reduce / comm: multi-GPU hooks (vaex_amd.dist)."""
        sa = self.sa
        if comm is None:
            comm = self.comm
        if isinstance(by, (list, tuple)):
            if len(by) > 1:
                return self._groupby_combined(list(by), agg_spec, reduce, comm, selection)
            by = by[0]
        if reduce is None and comm is not None:
            reduce = comm.allreduce
        key = self.columns[by]
        if np.ma.isMaskedArray(key):
            raise NotImplementedError("masked group keys")
        pf = _class_postfix(key)
        if pf.startswith("float") or pf.endswith("_non_native"):
            raise NotImplementedError("groupby on float / non-native keys")
        descs, names = [], []
        for name, d in agg_spec.items():
            names.append(name)
            if selection is not None:
                if d.selection is not None:
                    raise NotImplementedError("groupby: an aggregation with its own selection next to a filter")
                d = agg._Desc(d.name, d.column, selection)
            descs.append(d)
        if self.n == 0 and comm is None:
            return {by: np.array([], dtype=np.int64), **{n: np.array([]) for n in names}}
        # key range: one exact integer min/max pass (vxh_minmax_int); ranks agree on the global range
        self._prescan(by, key, pf, descs)
        kmin, kmax = self._key_range(by, key, pf)
        if comm is not None:
            kmin, kmax = comm.minmax(kmin, kmax)
        count = kmax - kmin + 1
        if 0 < count <= self.direct_groupby_cells:
            if count > self.dense_peel_cells and (comm is not None or self.n >= self.heavy_key_rows) and hasattr(sa, "groupby_run"):
                # a key range wider than one CU's LDS bins through the slab-partitioned pass: every row of one key goes to one slab's
                # queue — a heavy key overflows it (rows beyond its capacity are same-address device atomics) and is one workgroup's
                # work in pass 2.  Same remedy as in front of the fused hash pass: the heavy keys are peeled off.
                # Round 4: such a column goes through the FUSED hash pass with the heavy keys peeled inside it (one pass, ~1x the
                # uniform-key pass; the dense three-pass peel of round 3 costs ~4x) — when its sample shows heavy keys, on every rank
                # (the ranks must take the same branch: one MIN all-reduce), and the call is inside the fused pass's signature.
                if self.one_kernel_peel:
                    mine = self.n >= self.heavy_key_rows and self._heavy_keys(by, key, share=self.heavy_key_share_in_pass) is not None
                    skewed = mine if comm is None else not comm.all_agree(not mine)   # (any rank's sample: one MIN all-reduce)
                    if skewed:
                        fused = self._groupby_fused(by, pf, descs, names, comm, key_range=(kmin, kmax), filter_sel=selection)
                        if fused is not None:
                            self.last_groupby_info = dict(self.last_groupby_info or {}, dense_range_through_fused_pass=1)
                            return fused
                # (outside the fused pass's signature — min / max, integer value columns, several selections: round 3's three passes)
                peeled = self._groupby_dense_peeled(by, pf, key, descs, names, (kmin, kmax), comm, filter_sel=selection)
                if peeled is not None:
                    return peeled
            # Round 6: a key range wider than one workgroup's LDS, ONE float64 value column (or none), one GPU: the fused pass with a DIRECT
            # table — the record's remainder indexes gb_reduce's LDS accumulators, no keys, no probe (vxh_groupby.hip) — behind gb_scatter's
            # shared streams, instead of the slab-partitioned pair (part_scatter_f64 4.0 TB/s -> gb_scatter 5.0 TB/s of the same traffic)
            if (self.dense_through_fused and comm is None and count > self.dense_peel_cells and self.n >= self.heavy_key_rows and hasattr(sa, "groupby_run")
                    and sa.config_get("gb_direct") and sa.config_get("gb_compact")):   # (several value columns: one pass per column, _groupby_fused_split)
                fused = self._groupby_fused(by, pf, descs, names, comm, key_range=(kmin, kmax), filter_sel=selection)
                if fused is not None:
                    self.last_groupby_info = dict(self.last_groupby_info or {}, dense_range_through_fused_pass=1)
                    return fused
            # the key column bins itself (BinnerOrdinal with min_value): ONE pass, no hash map.  This is the
            # reference's dense-key simplification; for sparse keys in a small range it gives the same result
            # (empty cells are dropped below) without pass 1.
            binby = [dict(column=by, count=count, min_value=kmin)]
            if hasattr(sa, "finish"):
                # finishers + the drop of empty groups on the device: only the finished columns of the groups that exist
                # cross PCIe (vaex does this part with numpy on the full grids: vaex/agg.py:403-455, vaex/groupby.py:955-972)
                # which groups exist: count(*) > 0 — or the count of a value column the pass computes anyway, when that column
                # is known to hold no NaN (then the two counts are the same grid and the pass has one aggregator less: for
                # sum/mean/std of one column over 1e6 groups that is 128 instead of 256 slabs)
                present_desc = agg.count(selection=selection)
                if comm is None:
                    for d in descs:
                        if d.column is not None and _same_selection(d.selection, selection) and d.name in ("mean", "var", "std", "count") and not self._may_hold_nan(d.column):
                            present_desc = agg.count(d.column, selection=selection)
                            break
                specs, grid, aggs, want = self._pass(descs + [present_desc], binby, reduce=reduce)
                fin = [d.finish_spec(sa, [aggs[i] for i in ids]) for d, ids in zip(descs, want[:-1])]
                cols, index = sa.finish(fin, present=aggs[want[-1][0]], first=0, n=count, want_index=True)
                out_keys = np.asarray(index)
                if kmin:
                    out_keys += kmin  # (in place: the array is this call's own pinned buffer)
                vals = [np.asarray(c).astype(self._result_dtype(d), copy=False) for c, d in zip(cols, descs)]
                return {by: out_keys, **dict(zip(names, vals))}
            res = self._agg(descs + [agg.count(selection=selection)], binby=binby, edges=True, reduce=reduce)
            present = res[-1][:count] > 0
            out_keys = (np.arange(count, dtype=np.int64) + kmin)[present]
            vals = [r[:count][present] for r in res[:-1]]
            return {by: out_keys, **dict(zip(names, vals))}
        # scattered keys, plain count / sum / mean / var / std of float64 columns: ONE radix-partitioned pass with the hash
        # table probed in LDS (vxh_groupby_run) instead of the reference's two passes over a global table
        fused = self._groupby_fused(by, pf, descs, names, comm, key_range=(kmin, kmax) if self.n or comm is not None else None, filter_sel=selection)
        if fused is not None:
            return fused
        # pass 1: distinct keys on the GPU (ordered_set.update), united over ranks, sorted -> sealed map whose
        # ordinals are the rank of the key in ascending order (ordered_set::create): identical on every rank
        hm = getattr(sa, "ordered_set_" + pf)()
        if _is_device(key):
            hm.update(key)
        else:
            for i1 in range(0, self.n, 1 << 24):
                hm.update(np.ascontiguousarray(key[i1:i1 + (1 << 24)]))
        keys = np.array(hm.key_array())
        if comm is not None:
            keys = comm.union_keys(keys)
        out_keys = np.sort(keys)
        nuniq = len(out_keys)
        if nuniq == 0:
            return {by: out_keys, **{n: np.array([]) for n in names}}
        sealed = getattr(sa, "ordered_set_" + pf)(nuniq)
        sealed.set_keys(out_keys.astype(np.int64))
        res = self._agg_hash(descs + ([agg.count(selection=selection)] if selection is not None else []), by, pf, sealed, reduce, nuniq)
        vals = [r[1:1 + nuniq] for r in res] if res[0].shape[0] != nuniq else res
        if selection is not None:   # (the key set came from every row: groups without a row inside the filter do not exist)
            present = np.asarray(vals[-1]) > 0
            out_keys, vals = out_keys[present], [np.asarray(v)[present] for v in vals[:-1]]
        return {by: out_keys, **dict(zip(names, vals))}

    #: False: a filter over the groupby's own value columns goes through the keep-mask like any other (tests / timing comparison)
    groupby_fused_predicate = True

    def _groupby_pred_terms(self, selection, vcols):
        """([(value index, op, constant), ...], truth) when `selection` is a predicate of 1..4 plain terms `column <op> constant` whose
        columns are all among the first two float64 value columns `vcols` of the fused groupby — what vxh_groupby_run_selected evaluates
        inside gb_scatter — else None (arithmetic terms, other columns, a ready-made mask: the keep-mask road)."""
        if not self.groupby_fused_predicate or not getattr(self.sa, "GROUPBY_PRED", 0):
            return None
        try:
            sel = self._selection_mask(selection)
        except _predicate.Unsupported:
            return None
        if not isinstance(sel, _predicate.Predicate) or sel.programs or not 1 <= len(sel.terms) <= 4:
            return None
        terms = []
        for c, op, value in sel.terms:
            name = sel.columns[c]
            if name not in vcols[:2] or isinstance(value, bool):
                return None
            col = self.columns[name]
            if np.ma.isMaskedArray(col) or str(col.dtype).replace("torch.", "") != "float64":
                return None
            if isinstance(value, int) and abs(value) >= 2 ** 63:   # (the device selections take integer constants as int64 first: sel_eval's rule)
                return None
            terms.append((vcols.index(name), int(op), float(value)))
        return terms, int(sel.truth)

    def _groupby_fused(self, by, pf, descs, names, comm, key_range=None, filter_sel=None):
        """the vxh_groupby_run path, or None when the call is outside its signature (then: ordered_set + BinnerHash).

        filter_sel: the CALL's filter (groupby(selection=), a filtered vaex frame) — rows outside it reach nothing and groups without
        a row inside it do not exist: it is the pass's keep-mask.  A selection of the AGGREGATIONS' own (vaex.agg.count(selection=...);
        filter_sel None) is something else — the groups are those of ALL rows, and a group without a selected row reports count 0 /
        sum 0 / mean NaN (vaex/groupby.py:884-899) — so it must not become the keep-mask of the one pass (groups vanished).
        One selection shared by every aggregation runs the pass TWICE: all rows for the group keys, the kept rows for the values,
        left-joined on the sorted keys."""
        sa = self.sa
        if not hasattr(sa, "groupby_run") or self.n == 0 and comm is None:
            return None
        key = self.columns[by]
        vcols = []
        shared = descs[0].selection if descs else None   # ONE selection over the whole call: a keep-mask of the pass
        for d in descs:
            if d.name not in ("count", "sum", "mean", "var", "std") or not _same_selection(d.selection, shared):
                return None
            if d.column is not None and d.column not in vcols:
                vcols.append(d.column)
        # Round 6: several value columns.  The pass's fast forms carry ONE payload word per record (compact 10 / 12-byte records, the direct and
        # the tag table: 7.2 / 8.2 ms per 1e9 rows); two value columns ride 24-byte SoA records through the generic table (19.5 ms), three
        # or more did not ride at all.  Where the key range allows compact records the call is ONE PASS PER VALUE COLUMN (two columns: 14.4 /
        # 16.4 ms), elsewhere one pass per PAIR of columns; the passes see the same rows, so their sorted group keys are the same array.
        if len(vcols) >= 2:
            bits = 64 if key_range is None or key_range[0] > key_range[1] else (int(key_range[1]) - int(key_range[0])).bit_length()
            per_pass = 1 if bits <= self.compact_key_bits and sa.config_get("gb_compact") else 2
            if len(vcols) > per_pass:
                return self._groupby_fused_split(by, pf, descs, names, comm, key_range, filter_sel, vcols, per_pass)
        if shared is not None and not _same_selection(shared, filter_sel):
            if filter_sel is not None or pf != "int64":   # (an own selection next to a filter is refused in groupby(); the key pass lends the int64 key as payload)
                return None
            return self._groupby_fused_own_selection(by, pf, descs, names, comm, key_range, shared)
        values = []
        for c in vcols:
            col = self.columns[c]
            if np.ma.isMaskedArray(col) or str(col.dtype).replace("torch.", "") != "float64" or _is_device(col) != _is_device(key):
                return None
            values.append(col if _is_device(col) else np.ascontiguousarray(col))
        if not values:
            # only count(*) (value_counts, `n: count`): the pass needs a payload word per record — an int64 key column lends its own
            # 8 bytes (the pass's count / sum / sum2 of them are never read, only the row counts); other key widths: the general path
            if pf != "int64":
                return None
            vcols = ["__payload__"]
            values = [key if _is_device(key) else np.ascontiguousarray(key)]
        keep = None
        # Round 6 (late): a filter whose terms all read the pass's own float64 value columns (`df[df.v > 3].groupby(k, agg=sum(v))`) is evaluated
        # by gb_scatter on the payload words it loads anyway (vxh_groupby_run_selected) — no sel_eval pass in front, no keep byte per row.
        # (What the terms are depends on the call alone, not on a rank's rows: every rank takes the same branch.)
        pred = self._groupby_pred_terms(shared, vcols) if shared is not None and self.one_kernel_peel else None   # (the three-pass peel behind the knob works on masks)
        if shared is not None and pred is None:
            keep = self._mask_array(shared)
            usable = keep is not None and _is_device(keep) == _is_device(key)
            # (what a rank's mask turns out to be is rank-local; leaving alone here would strand the other ranks in the pass's
            #  collectives below — ADVICE r4: every rank-local exit is agreed on first)
            if not (usable if comm is None else comm.all_agree(usable)):
                return None
            keep = keep if _is_device(keep) else np.ascontiguousarray(_as_u8(keep))
        # heavy hitters (a default / missing-value key, the head of a Zipf law): every row of ONE key lands in ONE bucket of the
        # partitioned pass — the bucket's queue overflows its spare blocks from a ~25 % share on, and long before that its single
        # reduce workgroup is the whole pass (a 1 % key of 1e9 rows: 7 ms on one CU).  A sample of the keys finds them; they are
        # peeled off (`_groupby_peeled`): everything else takes the fused pass with the heavy rows masked out, the few heavy
        # keys are a dense groupby over their ordinals.
        # Round 4: the peel happens INSIDE the pass (vxh_groupby_run_peeled: gb_scatter looks every key up in an LDS copy of the heavy
        # list and adds such rows to per-workgroup partials instead of emitting a record) — no ordinal column, no keep-mask, no second
        # groupby over the heavy rows.  Row-sharded frames: every rank peels what is heavy in ITS shard; the partial groups of a heavy
        # key are ordinary partial groups in the cross-rank merge (a handful per key: nothing skewed about them).
        heavy = None
        if self.n >= self.heavy_key_rows and self.one_kernel_peel:
            heavy = self._heavy_keys(by, key, share=self.heavy_key_share_in_pass)
        elif (comm is not None or self.n >= self.heavy_key_rows) and pf in _PEEL_KEY_KINDS and not self.one_kernel_peel:
            # (the three-pass peel of round 3, kept behind the knob for the timing comparison: tools/r03_skew_groupby.py)
            try:
                import torch
            except ImportError:   # (the peel keeps its row-wise intermediates in torch tensors: no rank has it then)
                torch = None
            heavy3 = self._heavy_keys_all_ranks(by, key, comm) if torch is not None else None
            if heavy3 is not None:
                peeled = dk = dv = dkeep = None
                try:
                    # (host rows: the pass would copy them to the device anyway — here once, for both parts)
                    dev = lambda a: a if _is_device(a) else torch.from_numpy(np.ascontiguousarray(a)).cuda()
                    dk, dv, dkeep = dev(key), [dev(v) for v in values], None if keep is None else dev(keep)
                    ready = True
                except torch.cuda.OutOfMemoryError:   # (no room for the copies — the plain attempt below; every rank goes there together)
                    ready = False
                if ready if comm is None else comm.all_agree(ready):
                    peeled = self._groupby_peeled(by, pf, descs, names, vcols, dk, dv, dkeep, heavy3, comm=comm)
                dk = dv = dkeep = None
                if not _is_device(key) or peeled is None:
                    torch.cuda.empty_cache()   # (whole columns went through torch's allocator: the library's own hipMallocs need the room back)
                if peeled is not None:
                    return peeled
        res, failed, peeled_here = None, None, 0
        # the number of groups of an earlier call over the same key column (remembered like the key range): the pass sizes its
        # bucket tables for a known count at 80 % load instead of a guessed 2^20 at 50 % — half the buckets for 1e6 keys
        seen = self.__dict__.setdefault("_group_count_cache", {}).get(by)
        hint = int(seen[1]) if _memo_hit(seen, key) and keep is None and pred is None else 0
        if heavy is not None:
            hint = 0   # (skewed keys: as many buckets as the default gives — the light keys are still uneven, and a bucket is one workgroup's work)
        try:
            # (the measured key range: where it leaves a remainder of <= 32 bits below the bucket bits, the pass moves 12-byte records)
            kr = None if key_range is None or key_range[0] > key_range[1] else (int(key_range[0]), int(key_range[1]))
            res = (sa.groupby_run(key if _is_device(key) else np.ascontiguousarray(key), values, _DT_CODE[pf], key_range=kr, heavy=heavy, pred=pred) if pred is not None else
                   sa.groupby_run(key if _is_device(key) else np.ascontiguousarray(key), values, _DT_CODE[pf], keep=keep, key_range=kr, heavy=heavy) if keep is not None else
                   sa.groupby_run(key if _is_device(key) else np.ascontiguousarray(key), values, _DT_CODE[pf], hint, key_range=kr, heavy=heavy)) if self.n else None
            if res is not None and keep is None and pred is None:
                self.__dict__["_group_count_cache"][by] = _memo(key, len(res))
            peeled_here = int(res.info().get("heavy_keys_in_pass", 0)) if res is not None else 0
        except RuntimeError as e:
            if not str(e).startswith("groupby"):   # (anything the pass itself reports: too many / too skewed keys, no room for its queues)
                raise
            failed = e
        if comm is not None:
            # every rank takes the same branch: one rank's shard may be too skewed for the partitioned pass while the others'
            # are fine — then ALL of them answer through ordered_set + BinnerHash (whose collectives must match up)
            if not comm.all_agree(failed is None):
                return None
            try:
                res = self._groupby_fused_allranks(res, len(values), comm)
            except RuntimeError as e:
                if not str(e).startswith("groupby"):
                    raise
                failed = e
            if not comm.all_agree(failed is None):
                return None
        elif failed is not None:
            return None  # -> ordered_set + BinnerHash
        if res is None:   # (no rank holds a row)
            return {by: np.array([], dtype=np.int64), **{n: np.array([]) for n in names}}
        which = {"sum": sa.GB_SUM, "mean": sa.GB_MEAN, "var": sa.GB_VAR, "std": sa.GB_STD}
        out = {by: np.asarray(res.column(sa.GB_KEYS))}
        for name, d in zip(names, descs):
            if d.name == "count":
                out[name] = np.asarray(res.column(sa.GB_ROWS) if d.column is None else res.column(sa.GB_COUNT, vcols.index(d.column)))
            else:
                out[name] = np.asarray(res.column(which[d.name], vcols.index(d.column)))
        self.last_groupby_info = res.info()
        if pred is not None:
            self.last_groupby_info.update(selection_in_pass=len(pred[0]))   # (the filter's terms were evaluated by gb_scatter: no keep-mask)
        if heavy is not None:
            self.last_groupby_info.update(heavy_keys=peeled_here)   # (of THIS rank's pass; a cross-rank merge has none)
        return out

    def _groupby_fused_split(self, by, pf, descs, names, comm, key_range, filter_sel, vcols, per_pass):
        """_groupby_fused over more value columns than one pass carries: the aggregations are dealt to passes by their value column
        (`per_pass` columns each; count(*) rides the first), every pass is the whole fused groupby over the same rows and the same keep-mask,
        so the passes agree on the groups.  Collective under `comm`: the passes are entered by every rank in the same order (the split
        depends on the call's signature and the agreed key range only)."""
        groups = [vcols[i:i + per_pass] for i in range(0, len(vcols), per_pass)]
        out, infos = None, []
        for g, cols in enumerate(groups):
            picked = [(n, d) for n, d in zip(names, descs) if (d.column in cols) or (d.column is None and g == 0)]
            part = self._groupby_fused(by, pf, [d for _, d in picked], [n for n, _ in picked], comm, key_range=key_range, filter_sel=filter_sel)
            ok = part is not None and (out is None or np.array_equal(np.asarray(part[by]), np.asarray(out[by])))
            if not (ok if comm is None else comm.all_agree(ok)):
                return None
            infos.append(dict(self.last_groupby_info or {}))
            if out is None:
                out = {by: part[by]}
            out.update({n: part[n] for n, _ in picked})
        info = dict(infos[0])
        for k in ("ms_scatter", "ms_reduce", "ms_sort", "retries"):
            info[k] = sum(float(i.get(k, 0)) for i in infos)
        info.update(passes=len(groups), value_columns_per_pass=per_pass)
        self.last_groupby_info = info
        return {by: out[by], **{n: out[n] for n in names}}

    def _groupby_fused_own_selection(self, by, pf, descs, names, comm, key_range, shared):
        """aggregations sharing ONE selection of their own: the groups are those of all rows (pass 1: count(*) per key, no keep-mask),
        the values those of the kept rows (pass 2: the same call with the selection as its filter); groups pass 2 does not know get
        count 0 / sum 0 / mean, var, std NaN — what vaex's numpy finishers make of empty cells (vaex/agg.py:403-455).  Both passes are
        collectives under `comm`, entered by every rank in the same order (the branch depends on the call's signature only)."""
        keys_only = self._groupby_fused(by, pf, [agg.count()], ["__rows__"], comm, key_range=key_range, filter_sel=None)
        ok = keys_only is not None
        if comm is not None and not comm.all_agree(ok):
            return None
        if not ok:
            return None
        info_keys = dict(self.last_groupby_info or {})
        kept = self._groupby_fused(by, pf, descs, names, comm, key_range=key_range, filter_sel=shared)
        ok = kept is not None
        if comm is not None and not comm.all_agree(ok):
            return None
        if not ok:
            return None
        all_keys = np.asarray(keys_only[by])
        pos = np.searchsorted(all_keys, np.asarray(kept[by]))   # (both ascending; the kept groups are a subset)
        out = {by: all_keys}
        for name, d in zip(names, descs):
            col = np.asarray(kept[name])
            if d.name == "count":
                full = np.zeros(len(all_keys), dtype=col.dtype if len(col) else np.int64)
            elif d.name == "sum":
                full = np.zeros(len(all_keys), dtype=col.dtype if len(col) else np.float64)
            else:
                full = np.full(len(all_keys), np.nan, dtype=np.float64)
            full[pos] = col
            out[name] = full
        self.last_groupby_info = dict(self.last_groupby_info or {}, own_selection_two_passes=1, groups_of_all_rows=int(len(all_keys)),
                                      groups_with_a_kept_row=int(len(pos)), ms_keys_pass=info_keys.get("ms_scatter", 0) + info_keys.get("ms_reduce", 0))
        return out

    #: dense key ranges of 2^14 .. 2^21 cells take the fused pass with a direct LDS table (round 6; several value columns: one pass per column); False: the slab-partitioned BinnerOrdinal pass
    dense_through_fused = True
    #: key ranges of up to this many bits leave compact records in the fused pass whatever its bucket count (remainder <= 32 bits below the >= 8 bucket bits of a pass over an unknown or large number of groups): a call with
    #: several value columns is then one pass per column (_groupby_fused_split)
    compact_key_bits = 40
    #: heavy keys are peeled inside the fused pass (round 4); False: round 3's three passes (ordinals, keep-mask, a dense groupby of the heavy rows)
    one_kernel_peel = True
    #: rows from which a device-resident key column is sampled for heavy hitters before the fused hash groupby
    heavy_key_rows = 1 << 22
    #: sampled share from which a key counts as heavy (1/128: at most 128 of them)
    heavy_key_share = 1.0 / 128
    #: ... for the peel inside the fused pass: 1/1024 (128 sampled rows of 2^17; the 128 most frequent such keys) — a bucket is ONE reduce
    #: workgroup's work, and a key with 0.5 % of the rows still triples its bucket's share of a 512-bucket pass
    heavy_key_share_in_pass = 1.0 / 1024

    #: dense key ranges wider than this many cells are checked for heavy keys too (narrower ones live in one workgroup's LDS)
    dense_peel_cells = 1 << 14

    def _groupby_dense_peeled(self, by, pf, key, descs, names, key_range, comm=None, filter_sel=None):
        """the dense (BinnerOrdinal) groupby with the heavy keys peeled off, or None (no heavy key, or a call outside the peel's
        signature: aggregations other than count / sum / mean / var / std / min / max over plain columns, selections that differ)"""
        if pf not in _PEEL_KEY_KINDS or not descs:
            return None
        shared = descs[0].selection
        if not _same_selection(shared, filter_sel):
            return None   # (the aggregations' own selection is not a filter: the peel's keep-mask would drop the groups without a kept row — the plain dense pass keeps them)
        cols = []
        for d in descs:
            if d.name not in ("count", "sum", "mean", "var", "std", "min", "max") or not _same_selection(d.selection, shared):
                return None
            if d.column is not None and d.column not in cols:
                if np.ma.isMaskedArray(self.columns[d.column]) or "_non_native" in _class_postfix(self.columns[d.column]):
                    return None
                cols.append(d.column)
        try:
            import torch
        except ImportError:
            return None
        heavy = self._heavy_keys_all_ranks(by, key, comm)   # (the signature checks above come out the same on every rank: a collective from here on)
        if heavy is None:
            return None
        try:
            # rank-local preparation (uploads: a rank may run out of memory alone) — then ALL ranks decide together whether the peel
            # goes ahead: its sub-frames' groupbys and the heavy part's all-reduce are collectives
            dkey = keep = dvals = None
            try:
                dev = lambda a: a if _is_device(a) else torch.from_numpy(np.ascontiguousarray(a)).cuda()
                if shared is not None:
                    keep = self._mask_array(shared)
                    keep = dev(keep if _is_device(keep) else _as_u8(keep))
                dkey, dvals = dev(key), [dev(self.columns[c]) for c in cols]
                ready = True
            except torch.cuda.OutOfMemoryError:
                ready = False
            if not (ready if comm is None else comm.all_agree(ready)):
                return None
            return self._groupby_peeled(by, pf, descs, names, cols, dkey, dvals, keep, heavy, dense_range=key_range, comm=comm)
        finally:
            dkey = keep = dvals = None
            if not _is_device(key):
                torch.cuda.empty_cache()   # (whole columns went through torch's allocator: the library's own hipMallocs need the room back)

    def _heavy_keys_all_ranks(self, by, key, comm):
        """the heavy keys of this rank's sample, or — row-sharded frames — the sorted union over the ranks (one `union_keys`
        exchange of <= 128 keys per rank: every rank peels the same set, whether or not a key is heavy in its own shard; round 4).
        None when no rank found one.  Every rank must call this (it is a collective under `comm`)."""
        mine = self._heavy_keys(by, key) if self.n >= self.heavy_key_rows else None
        if comm is None or comm.world() == 1:
            return mine
        allk = np.asarray(comm.union_keys(np.zeros(0, dtype=np.int64) if mine is None else np.asarray(mine, dtype=np.int64)), dtype=np.int64)
        return allk if len(allk) else None

    def _heavy_keys(self, by, key, share=None):
        """keys holding >= `share` (default heavy_key_share) of a strided sample of 2^17 rows of the key column — the 128 most frequent
        of them at most (ascending int64 array) — or None.  A heuristic: correctness never rests on it (a missed heavy key only costs
        time, a false one a little).  Remembered per column object like the key range."""
        share = self.heavy_key_share if share is None else share
        cache = self.__dict__.setdefault("_heavy_cache", {})
        hit = cache.get((by, share))
        if _memo_hit(hit, key):
            return hit[1]
        m = 1 << 17
        step = max(1, self.n // m)
        if isinstance(key, np.ndarray):
            uniq, cnt = np.unique(np.asarray(key[::step][:m]), return_counts=True)
            sel = cnt >= max(8, int(min(m, len(key[::step])) * share))
            uniq, cnt = uniq[sel], cnt[sel]
            hv = uniq[np.argsort(-cnt, kind="stable")[:128]]
        elif hasattr(self.sa, "sample_heavy_keys") and _is_device(key) and _class_postfix(key) in _DT_CODE and not _class_postfix(key).startswith("float"):
            # the library's own sampler (round 6: gather + rocPRIM sort + run-length encode; until then torch.unique — whose kernels torch loads
            # on first use, ~100 ms of a process's first groupby)
            m_eff = min(m, -(-self.n // step))
            hv = np.asarray(self.sa.sample_heavy_keys(key, _DT_CODE[_class_postfix(key)], m, max(8, int(m_eff * share)), 128))
            heavy = hv.astype(np.int64) if len(hv) else None
            cache[(by, share)] = _memo(key, heavy)
            return heavy
        else:
            try:
                import torch
            except ImportError:
                return None
            if not isinstance(key, torch.Tensor):
                return None
            sample = key[::step][:m]
            try:
                uniq, cnt = torch.unique(sample, return_counts=True)
            except RuntimeError:   # (a dtype torch does not sort)
                cache[(by, share)] = _memo(key, None)
                return None
            sel = cnt >= max(8, int(len(sample) * share))
            uniq, cnt = uniq[sel].cpu().numpy(), cnt[sel].cpu().numpy()
            hv = uniq[np.argsort(-cnt, kind="stable")[:128]]
        heavy = np.sort(hv.astype(np.int64)) if len(hv) else None
        cache[(by, share)] = _memo(key, heavy)
        return heavy

    def _groupby_peeled(self, by, pf, descs, names, vcols, key, values, keep, heavy, dense_range=None, comm=None):
        """the fused hash groupby with the rows of the `heavy` keys taken out of it: a sealed device set of the heavy keys maps
        every row to the key's ordinal or -1 (vxh_hashmap_map_ordinal: a table of <= 128 keys stays in the caches); rows with -1
        take the partitioned pass (keep-mask), the others a dense groupby over the ordinals — <= 128 groups, LDS-resident, at the
        rate of a binned count.  The two group sets are disjoint; the result is their union in ascending key order.
        None: the light part is still too much for the partitioned pass (-> ordered_set + BinnerHash).
        comm (round 4): row-sharded frames — `heavy` is the set all ranks agreed on; the light part's partial groups are merged across
        the ranks like the fused pass's (all-gather + MERGE kernels), the heavy part is a dense groupby whose grids are all-reduced;
        every rank returns the whole table's groups."""
        import torch
        sa = self.sa
        ords = ords_dev = light = None
        try:   # (rank-local: 9 more bytes per row — a rank short of memory takes EVERY rank out of the peel, before its first collective)
            sealed = getattr(sa, "ordered_set_" + pf)(len(heavy))
            sealed.set_keys(heavy)
            ords_dev = sealed.map_ordinal_device(key)
            ords = torch.as_tensor(ords_dev, device="cuda")          # int64: rank of the row's key among the heavy ones, -1 = not heavy
            light = ords < 0
            if keep is not None:   # (1 = keep, anything else drops the row: src/agg_count.cpp:50)
                if not isinstance(keep, torch.Tensor):
                    keep = torch.as_tensor(keep, device="cuda")
                light = light & ((keep.view(torch.uint8) if keep.dtype == torch.bool else keep) == 1)
            light = light.to(torch.uint8)
            ready = True
        except (torch.cuda.OutOfMemoryError, MemoryError):
            ready = False
        except RuntimeError as e:
            if "memory" not in str(e).lower():
                raise
            ready = False
        if not (ready if comm is None or comm.world() == 1 else comm.all_agree(ready)):
            return None
        plain = {n: agg._Desc(d.name, d.column, None) for n, d in zip(names, descs)}
        if dense_range is not None:
            # the light rows of a dense key range: the same dense groupby with the heavy rows masked out
            fl = Frame({by: key, **dict(zip(vcols, values)), "__light__": light}, chunk_size=self.chunk_size, nthreads=self.nthreads, superagg=sa, comm=comm)
            fl.heavy_key_rows = 1 << 62
            fl.direct_groupby_cells = self.direct_groupby_cells
            fl.__dict__["_key_range_cache"] = {by: _memo(key, dense_range)}
            out = fl.groupby(by, plain, selection="__light__")
            info = {"dense": 1}
        else:
            res, failed = None, False
            try:
                res = sa.groupby_run(key, values, _DT_CODE[pf], keep=light) if len(key) else None
            except RuntimeError as e:
                if not str(e).startswith("groupby"):
                    raise
                failed = True
            if comm is not None and comm.world() > 1:
                if not comm.all_agree(not failed):   # (every rank leaves the peel together)
                    return None
                try:
                    res = self._groupby_fused_allranks(res, len(values), comm)
                except RuntimeError as e:
                    if not str(e).startswith("groupby"):
                        raise
                    failed = True
                if not comm.all_agree(not failed):
                    return None
            elif failed:
                return None
            which = {"sum": sa.GB_SUM, "mean": sa.GB_MEAN, "var": sa.GB_VAR, "std": sa.GB_STD}
            if res is None:   # (no rank holds a light row)
                out = {by: np.zeros(0, dtype=np.int64), **{n: np.zeros(0, dtype=np.int64 if d.name == "count" else np.float64) for n, d in zip(names, descs)}}
                info = {}
            else:
                out = {by: np.asarray(res.column(sa.GB_KEYS))}
                for name, d in zip(names, descs):
                    if d.name == "count":
                        out[name] = np.asarray(res.column(sa.GB_ROWS) if d.column is None else res.column(sa.GB_COUNT, vcols.index(d.column)))
                    else:
                        out[name] = np.asarray(res.column(which[d.name], vcols.index(d.column)))
                info = dict(res.info())
        # the heavy keys: their ordinals are a dense key column
        sub = {"__heavy__": ords}
        for c, col in zip(vcols, values):
            sub[c] = col
        selection = None
        if keep is not None:
            sub["__keep__"] = keep
            selection = "__keep__"
        f = Frame(sub, chunk_size=self.chunk_size, nthreads=self.nthreads, superagg=sa, comm=comm)
        f.heavy_key_rows = 1 << 62
        # what the sub-frame would scan the rows for is known: the ordinals' range, and whether a value column holds NaN (as far as
        # this frame has looked: the columns are the same objects)
        f.__dict__["_key_range_cache"] = {"__heavy__": _memo(ords, (-1, len(heavy) - 1))}
        nan_seen = self.__dict__.get("_nan_cache", {})
        f.__dict__["_nan_cache"] = {c: _memo(col, nan_seen[c][1]) for c, col in zip(vcols, values) if _memo_hit(nan_seen.get(c), col)}
        hres = f.groupby("__heavy__", plain, selection=selection)
        for c, col in zip(vcols, values):   # (what the sub-frame learnt about NaN in a column of this frame is this frame's to keep)
            hit = f.__dict__.get("_nan_cache", {}).get(c)
            if hit is not None and self.columns.get(c) is col:
                self.__dict__.setdefault("_nan_cache", {})[c] = hit
        hk = np.asarray(hres.pop("__heavy__")).astype(np.int64)
        inside = hk >= 0                                          # (the group of ordinal -1 is every light row: not a group)
        hkeys = heavy[hk[inside]]
        keys = np.concatenate([np.asarray(out[by]).astype(np.int64), hkeys])
        order = np.argsort(keys, kind="stable")
        merged = {by: keys[order].astype(np.asarray(out[by]).dtype, copy=False)}
        for n in names:
            a, b = np.asarray(out[n]), np.asarray(hres[n])[inside]
            merged[n] = np.concatenate([a, b.astype(a.dtype, copy=False)])[order]
        info.update(heavy_keys=int(len(heavy)), heavy_groups=int(inside.sum()))
        self.last_groupby_info = info
        del ords, ords_dev
        return merged

    def _groupby_fused_allranks(self, res, nv, comm):
        """every rank's partial groups -> the groups of all ranks' rows: the partial columns {key, rows, count_j, sum_j, sum2_j}
        are all-gathered as tensors of the backend's device (dist.Comm.all_gather_arrays: RCCL over xGMI on GPUs) and merged
        by the same two kernels in MERGE mode on every rank (the cross-rank form of the reference's per-thread merge,
        vaex/cpu.py:788-796).  A rank without rows contributes empty columns."""
        sa = self.sa
        if comm.world() == 1:
            return res
        i8, f8 = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.float64)
        mine = [np.asarray(res.column(sa.GB_KEYS)) if res is not None else i8, np.asarray(res.column(sa.GB_ROWS)) if res is not None else i8]
        for j in range(nv):
            mine += [np.asarray(res.column(c, j)) if res is not None else (i8 if c == sa.GB_COUNT else f8) for c in (sa.GB_COUNT, sa.GB_SUM, sa.GB_SUM2)]
        parts = comm.all_gather_arrays(mine)
        cat = lambda i: np.concatenate([p[i] for p in parts])
        keys = cat(0)
        if len(keys) == 0:
            return None
        return sa.groupby_merge(keys, cat(1), [cat(2 + 3 * j) for j in range(nv)], [cat(3 + 3 * j) for j in range(nv)], [cat(4 + 3 * j) for j in range(nv)])

    def _agg_hash(self, descs, by, pf, hm, reduce, nuniq=None):
        """Same fused pass with a BinnerHash on the key column (cells: [unknown, ordinal 0..N-1, null])."""
        sa = self.sa
        prims, index, want = [], {}, []
        for d in descs:
            ids = []
            for p in d.prims():
                k = p.key()
                if k not in index:
                    index[k] = len(prims)
                    prims.append(p)
                ids.append(index[k])
            want.append(ids)
        # build the pass by hand: one hash binner
        key = self.columns[by]
        device = _is_device(key)
        slots = 1 if device and all(p.column is None or _is_device(self._col(p.column, p.as_float64)) for p in prims) else max(1, self.nthreads)
        binner = getattr(sa, "BinnerHash_" + pf)(slots, by, hm)
        grid = sa.Grid([binner])
        aggs = []
        for p in prims:
            cpf = "int64" if p.column is None else _class_postfix(self._col(p.column, p.as_float64))
            cls = getattr(sa, _KIND_CLASS[p.kind] + cpf)
            aggs.append(cls(grid, slots, slots, p.moment) if p.kind == "summoment" else cls(grid, slots, slots))
        step = self.n if slots == 1 and device else self.chunk_size
        slot = 0
        for i1 in range(0, self.n, max(1, step)):
            i2 = min(self.n, i1 + step)
            refs = []
            k = key[i1:i2] if device else np.ascontiguousarray(key[i1:i2])
            binner.set_data(slot, k); binner.clear_data_mask(slot); refs.append(k)
            for a, p in zip(aggs, prims):
                if p.column is not None:
                    c = self._col(p.column, p.as_float64)
                    d = c[i1:i2] if _is_device(c) else np.ascontiguousarray(c[i1:i2])
                    a.set_data(slot, d, 0); refs.append(d)
                sel = self._mask_array(p.selection)
                if sel is not None:
                    m = _as_u8(sel[i1:i2]); a.set_data_mask(slot, m); refs.append(m)
                else:
                    a.clear_data_mask(slot)
            grid.bin(slot, aggs, i2 - i1)
            slot = (slot + 1) % slots
        if reduce is not None:
            from .dist import with_reduced
            aggs = with_reduced(aggs, reduce(aggs))
        if hasattr(sa, "finish") and nuniq is not None:  # cells [unknown, ordinal 0..N-1, null]: finish the N group cells on the device
            fin = [d.finish_spec(sa, [aggs[i] for i in ids]) for d, ids in zip(descs, want)]
            cols, _ = sa.finish(fin, present=None, first=1, n=nuniq, want_index=False)
            return [np.asarray(c).astype(self._result_dtype(d), copy=False) for c, d in zip(cols, descs)]
        raw = [np.asarray(a.get_result()) for a in aggs]  # (get_result already returns a fresh array)
        return [d.finish([raw[i] for i in ids]) for d, ids in zip(descs, want)]


_DT_CODE = {"float64": 0, "float32": 1, "int64": 2, "int32": 3, "int16": 4, "int8": 5, "uint64": 6, "uint32": 7, "uint16": 8, "uint8": 9, "bool": 10}
