"""df.groupby(<integer key columns>, agg=...) of an unmodified vaex on the device groupby.

vaex answers a groupby in two passes over the rows plus numpy finishers (vaex/groupby.py:602-1017): the distinct keys go
through `ordered_set` (vaex/hash.py:152-214, vaex/cpu.py:285-404), every chunk's keys are mapped to ordinals on the host
(`_ordinal_values`, vaex/groupby.py:303-317), the ordinals are binned by BinnerOrdinal with one aggregator per primitive,
and mean / var / std are finished with numpy over the full grids (vaex/agg.py:386-455).  vaex_amd.install() swaps the HIP
classes under that machinery already; this module goes one step further for the calls the device groupby of
vaex_amd.binned.Frame covers — it answers `DataFrame.groupby(by, agg=...)` itself:

    keys   1..8 real integer columns (bool, int8 .. int64, uint8 .. uint32; numpy / memory-mapped, no missing values)
    agg    count(*) / count(x) / sum(x) / mean(x) / var(x) / std(x) / min(x) / max(x) on real numeric columns without missing values,
           each with or without a selection of vaex_amd.predicate's subset (an expression or a named selection: the groups are those of
           ALL rows, an aggregation sees the rows its selection keeps — vaex/groupby.py:884-899) — given as vaex.agg objects, names
           ('count', 'mean', ...), lists or {name: ...} dicts, i.e. every form GroupByBase._agg accepts: the actions are walked by
           vaex's OWN loop (vaex/groupby.py:688-745) over a frame that records the aggregations instead of running them
    frame  unfiltered, or filtered by comparison expressions over real numeric columns (vaex_amd.predicate's subset: the filter
           becomes a device predicate); row_limit=None

and builds the resulting DataFrame the way GroupBy.agg does (vaex/groupby.py:955-983): one row per group that exists, the key
columns first.  Dense key ranges bin themselves in ONE partitioned pass (BinnerOrdinal(min_value) + vxh_finish: what vaex
reaches after its distinct-key pass through the BinnerInteger simplification, vaex/groupby.py:263-272), scattered keys go
through the fused radix-partitioned hash aggregation (vxh_groupby_run), several keys are packed into one on the device
(vxh_pack_keys: vaex's GrouperCombined, vaex/groupby.py:526-584).  Anything outside that signature — and any failure the
device path reports — falls through to vaex's own groupby, which then still runs on the HIP classes task by task.

`delay=True` (round 5): the same signature as a TASK of the executor's pass — TaskGroupbyHip below: the executor's chunks are appended to the
plan's columns in HBM, the fused groupby runs when the pass is over; one pass for any number of delayed groupbys and the caller's other tasks.

Group order: ascending by key(s) (descending with sort=True, ascending=False).  vaex's own order without `sort` is its hash
set's insertion order — unspecified; with sort=True it is this one.  The key column comes back the way vaex types it: a
masked int64 array without masked entries when vaex would have simplified to BinnerInteger (key range <= 4/3 of the distinct
keys), else the narrowest signed integer type that holds the key range (vaex/groupby.py:263-277).
"""
import collections.abc
import os
import itertools
import threading
import weakref

import numpy as np

from . import binned

#: what the most recent DataFrame.groupby(..., agg=...) ran on: {"path": "device" | "vaex", "kernel": ..., "why": ...}
last = {}
#: df.groupby calls answered by the device groupby / handed on to vaex's own two passes (with the reasons)
stats = {"device": 0, "task": 0, "vaex": 0, "why": {}}

_KEY_KINDS = ("bool", "int8", "int16", "int32", "int64", "uint8", "uint16", "uint32")
_TINY_KEYS = ("bool", "int8", "uint8")   # vaex bins these with BinnerInteger straight away (vaex/groupby.py:593-595): no combined grouper, own key typing
_VALUE_KINDS = ("float64", "float32", "int64", "int32", "int16", "int8", "uint32", "uint16", "uint8")
_AGG_NAMES = {"AggCount": "count", "AggSum": "sum", "AggMin": "min", "AggMax": "max"}


class _Decline(Exception):
    """the call is outside the device groupby's signature: vaex's own code answers it"""


class _Streamed:
    """a real numeric column that is not one numpy array — a pyarrow Array / ChunkedArray without nulls (arrow / parquet files,
    vaex/arrow/dataset.py), a dataset's ColumnProxy (sliced / concatenated / renamed datasets, vaex/dataset.py:575-611): the plan knows its
    dtype and length, the rows come chunk by chunk through the executor (the groupby TASK: TaskGroupbyHip) — round 6"""

    def __init__(self, dtype, n, source):
        self.dtype, self.n, self.source = np.dtype(dtype), int(n), source

    def __len__(self):
        return self.n


def _streamed_dtype(df, name, ar):
    """numpy dtype of a column object whose rows can be streamed to the device groupby as plain numeric chunks, else None"""
    from . import predicate
    dt = predicate.plain_numeric_dtype(ar)          # arrow without nulls (and plain numpy)
    if dt is not None:
        return dt
    if type(ar).__name__ == "ColumnProxy" and hasattr(ar, "ds"):
        try:
            dt = df.data_type(name)
            if dt.is_numeric or dt == bool:         # (missing values show when the chunks arrive: the task then leaves the pass to vaex)
                dt = np.dtype(dt.numpy)
                return dt if dt.isnative else None
        except Exception:   # noqa: BLE001  (a column type vaex itself cannot describe as numpy)
            return None
    return None


def _real_column(df, expression, kinds, what, materialise=True):
    """the numpy array behind `expression` when it names a real, unmasked column of one of `kinds` (active range applied) — or a _Streamed
    stand-in for a real numeric column held in another container (its rows then reach the device through the executor's chunks)"""
    name = str(expression)
    if name not in df.columns:
        return _virtual_column(df, name, kinds, what, materialise)
    ar = df.columns[name]
    i1, i2 = df._index_start, df._index_end
    if np.ma.isMaskedArray(ar) or not isinstance(ar, np.ndarray):
        kind = "masked numpy" if np.ma.isMaskedArray(ar) else f"{type(ar).__module__.split('.')[0]}.{type(ar).__name__}" + (f"[{ar.type}]" if hasattr(ar, "type") and hasattr(ar, "null_count") else "")
        dt = None if np.ma.isMaskedArray(ar) else _streamed_dtype(df, name, ar)
        if dt is None:
            raise _Decline(f"{what} {name!r} is not a plain numpy column ({kind})")
        if dt.name not in kinds:
            raise _Decline(f"{what} {name!r} has dtype {dt}")
        n = len(ar)
        return name, _Streamed(dt, (n if i2 is None else i2) - i1, ar)
    if ar.dtype.name not in kinds or not ar.dtype.isnative or ar.ndim != 1:
        raise _Decline(f"{what} {name!r} has dtype {ar.dtype}")
    if i2 is None:
        i2 = len(ar)
    if i1 != 0 or i2 != len(ar):
        ar = ar[i1:i2]
    return name, ar


#: a virtual column / expression is materialised on the host for the device groupby up to this many rows (one numpy array of the frame's length)
materialise_max_rows = 1 << 27


def _virtual_column(df, name, kinds, what, materialise=True):
    """`name` is a virtual column or an expression (round 6).  An alias of a real column (df['long_name'] = df.x: vaex/dataframe.py:3596-3640 stores
    the expression 'x') is that column; anything else is evaluated ONCE on the host by vaex itself (df.evaluate over the active range,
    unfiltered: the frame's filter is applied by the groupby) when the frame is small enough for one array — vaex's own passes evaluate it too,
    chunk by chunk, twice for a key.  Masked results, strings and other dtypes decline."""
    virtual = getattr(df, "virtual_columns", {})
    label = getattr(df[name], "_label", name) if name in virtual else name
    seen = set()
    target = name
    while target in virtual and target not in seen:   # an alias chain ends at a real column
        seen.add(target)
        target = str(virtual[target]).strip()
    if target in df.columns and target != name:
        return name, _real_column(df, target, kinds, what)[1]
    if not materialise or len(df) > materialise_max_rows or df.length_original() > materialise_max_rows:
        raise _Decline(f"{what} {label!r} is not a real column")
    try:
        ar = df.evaluate(name, filtered=False, parallel=False)
    except Exception as e:   # noqa: BLE001  (an expression vaex cannot evaluate to an array: its own groupby says so)
        raise _Decline(f"{what} {label!r} is not a real column ({type(e).__name__})")
    if np.ma.isMaskedArray(ar) or not isinstance(ar, np.ndarray) or ar.ndim != 1:
        raise _Decline(f"{what} {label!r} is not a real column (evaluates to {type(ar).__name__})")
    if ar.dtype.name not in kinds or not ar.dtype.isnative:
        raise _Decline(f"{what} {label!r} has dtype {ar.dtype}")
    return name, np.ascontiguousarray(ar)


def _binner_object_key(df, b, n_keys, run_pending=True):
    """a binner OBJECT passed as a key (round 6) -> (key column expression, what the result must look like), or _Decline.
    vaex.groupby.Grouper(expression, sort=, ascending=) over an integer column (vaex/groupby.py:226-330): the object has run its distinct-key
    pass already when it was made; the groups are its bin_values IN ITS ORDER (sorted either way, or the hash map's own), typed the narrowest
    signed integer that holds them (no BinnerInteger simplification: allow_simplify is the wrapper's own, :599).
    vaex.groupby.BinnerInteger(expression) over bool / int8 / uint8 (:147-205): what df.groupby(<such a column>) makes itself (:593-596)."""
    import vaex.array_types
    import vaex.groupby
    kind = type(b).__name__
    if n_keys != 1:
        raise _Decline(f"binner object as key ({kind} next to other keys)")   # (several binner objects: the combine decision and the cell order are the objects' own)
    if type(b) is vaex.groupby.Grouper:
        if not hasattr(b, "hashmap_unique") and run_pending and getattr(getattr(b, "_promise", None), "isPending", False):
            # the object's distinct-key pass is scheduled, not run (vaex/groupby.py:298: delay=True; GroupBy.__init__ would run it now, :1021)
            b.df.execute()
        if getattr(b, "simpler", None) is not None or not hasattr(b, "bin_values") or not hasattr(b, "hashmap_unique"):
            raise _Decline(f"binner object as key ({kind} that has not run its distinct-key pass)")
        if b.df.dataset != df.dataset:
            raise _Decline(f"binner object as key ({kind} of another dataset)")
        bv = b.bin_values
        if hasattr(bv, "null_count"):   # (an arrow array: the sorted forms come back through pyarrow)
            if bv.null_count:
                raise _Decline(f"binner object as key ({kind} with a missing-value group)")
            bv = vaex.array_types.to_numpy(bv)
        if np.ma.isMaskedArray(bv) or not isinstance(bv, np.ndarray) or bv.dtype.kind not in "iu" or getattr(b.hashmap_unique, "has_null", False) or getattr(b.hashmap_unique, "has_nan", False):
            raise _Decline(f"binner object as key ({kind} over {getattr(bv, 'dtype', type(bv).__name__)})")
        return str(b.expression), {"kind": "grouper", "bin_values": bv}
    if type(b) is vaex.groupby.BinnerInteger:
        if b.dtype.numpy.name not in _TINY_KEYS or getattr(b, "dropmissing", False) or b.df.dataset != df.dataset:
            raise _Decline(f"binner object as key ({kind})")
        return str(b.expression), {"kind": "integer", "invert": bool(b.invert)}
    raise _Decline(f"binner object as key ({kind})")


class _RecordingFrame:
    """the DataFrame as GroupByBase._agg sees it: `_agg` records the aggregation instead of scheduling it, the rest is the real frame"""

    def __init__(self, df):
        self.__dict__["_df"] = df
        self.__dict__["recorded"] = []

    def _agg(self, aggregate, binners=(), delay=False, progress=None, **kw):
        self.recorded.append(aggregate)
        return None

    def __getattr__(self, name):
        return getattr(self._df, name)

    def __getitem__(self, item):
        return self._df[item]


def _normalise_actions(df, keys, actions):
    """[(output column name, vaex aggregator descriptor)] in vaex's order, by running vaex's own action loop — GroupByBase._agg
    (vaex/groupby.py:688-745: lists, dicts, names, callables over all columns, override names) — on a bare GroupByBase whose
    frame records what it is asked to aggregate"""
    import vaex.agg
    import vaex.groupby
    rec = _RecordingFrame(df)
    shell = object.__new__(vaex.groupby.GroupByBase)
    shell.df = rec
    shell.binners = ()
    shell.groupby_expression = list(keys)
    for a in ([actions] if isinstance(actions, (str, vaex.agg.AggregatorDescriptor)) or not isinstance(actions, collections.abc.Iterable) else
              (actions.values() if isinstance(actions, collections.abc.Mapping) else actions)):
        for one in (a if isinstance(a, (list, tuple)) else [a]):
            if isinstance(one, str) and one != "count" and one not in vaex.agg.aggregates:
                raise _Decline(f"unknown aggregate {one!r}")
    try:
        grids = vaex.groupby.GroupByBase._agg(shell, actions, None)
    except (KeyError, TypeError, ValueError, AttributeError) as e:
        raise _Decline(f"actions vaex does not take: {type(e).__name__}: {e}")
    if len(grids) != len(rec.recorded):
        raise _Decline("duplicate output column")
    for a in rec.recorded:   # (vaex's loop sets it on every aggregation it schedules; these objects are the caller's and may be reused)
        a.edges = False
    return list(zip(grids.keys(), rec.recorded))


def _expression_types():
    import vaex.agg
    return (vaex.agg.AggregatorExpressionUnary, vaex.agg.AggregatorExpressionBinary, vaex.agg.AggregatorExpressionBinaryScalar)


def _translate_tree(df, aggregate, columns, predicates, spec):
    """an aggregator expression -> a tree of ("leaf", hidden output name) / ("op", callable, children...); the leaves are entered into `spec`"""
    import vaex.agg
    if isinstance(aggregate, vaex.agg.AggregatorExpressionUnary):
        return ("op", aggregate.finish, _translate_tree(df, aggregate.agg, columns, predicates, spec))
    if isinstance(aggregate, vaex.agg.AggregatorExpressionBinary):
        return ("op", aggregate.finish, _translate_tree(df, aggregate.agg1, columns, predicates, spec), _translate_tree(df, aggregate.agg2, columns, predicates, spec))
    if isinstance(aggregate, vaex.agg.AggregatorExpressionBinaryScalar):
        return ("op", aggregate.finish, _translate_tree(df, aggregate.agg, columns, predicates, spec))
    name = f"__leaf_{len(spec)}"
    spec[name] = _translate(df, aggregate, columns, predicates)
    return ("leaf", name)


def _finish_tree(tree, res):
    if tree[0] == "leaf":
        return np.asarray(res[tree[1]])
    with np.errstate(all="ignore"):
        return np.asarray(tree[1](*[_finish_tree(t, res) for t in tree[2:]]))


def _translate(df, aggregate, columns, predicates):
    """vaex aggregator descriptor -> binned.agg descriptor; the value column is entered into `columns`, the compiled selection into `predicates`"""
    import vaex.agg
    selection = _selection_of(df, aggregate, columns, predicates)
    if isinstance(aggregate, vaex.agg.AggregatorDescriptorBasic):
        kind = _AGG_NAMES.get(aggregate.name)
        if kind is None or aggregate.agg_args:
            raise _Decline(f"aggregator {aggregate.name}")
    elif type(aggregate) is vaex.agg.AggregatorDescriptorMean:
        kind = "mean"
    elif type(aggregate) in (vaex.agg.AggregatorDescriptorVar, vaex.agg.AggregatorDescriptorStd):
        kind = "std" if isinstance(aggregate, vaex.agg.AggregatorDescriptorStd) else "var"  # (ddof never enters vaex's formula: vaex/agg.py:440-455)
    else:
        raise _Decline(f"aggregator {type(aggregate).__name__}")
    expressions = list(aggregate.expressions)
    if kind == "count" and not expressions:
        return binned.agg.count(selection=selection)
    if len(expressions) != 1:
        raise _Decline("aggregator over several expressions")
    name, ar = _real_column(df, expressions[0], _VALUE_KINDS, "aggregated expression")
    if selection is not None and kind in ("min", "max"):
        raise _Decline("min / max with a selection")   # (a group without a selected row: vaex hands back the dtype's extreme, masked or not by dtype)
    columns[name] = ar   # (var / std of an integer column: the primitives run on astype(float64), as vaex/agg.py:427 does)
    return getattr(binned.agg, kind)(name, selection=selection)


def _selection_of(df, aggregate, columns, predicates):
    """the aggregation's selection as an expression of the device predicate subset (its columns entered into `columns`), or None.
    The Predicate compiled HERE — against the frame's real and virtual columns — is entered into `predicates` under the expression: the
    binned.Frame that runs the call is seeded with it (_run), since it knows the plan's real columns only and could not inline `r < 1`
    over a virtual column `r` by itself."""
    sel = getattr(aggregate, "selection", None)
    if sel is None or sel is False:
        return None
    if isinstance(sel, (list, tuple)):
        raise _Decline("aggregation with several selections")
    if df.filtered:
        raise _Decline("aggregation with a selection on a filtered frame")
    from . import vaex_selection, predicate
    sel = "default" if sel is True else str(sel)
    if df.has_selection(sel):   # a named selection: its history as one expression
        sel = vaex_selection.named_expression(df, sel)
        if not sel:
            raise _Decline("named selection outside the device predicate subset")
    try:
        pred = predicate.compile_selection(sel, vaex_selection._known_columns(df), virtual=vaex_selection._virtual_columns(df))
    except predicate.Unsupported as e:
        raise _Decline(f"selection outside the device predicate subset ({e})")
    for c in pred.columns:
        if c not in columns:
            columns[c] = _real_column(df, c, tuple(k for k in vaex_selection._NUMERIC), "selection column")[1]
    predicates[sel] = pred
    return sel


def _key_column_like_vaex(values, source_kind=None):
    """a key column of the result typed the way vaex's groupers hand it back (vaex/groupby.py:147-205, :263-277) — decided per key
    from its distinct values: BinnerInteger (range <= 4/3 of the distinct keys) -> int64 with an (empty) mask; else Grouper -> the
    narrowest signed integer type that holds the range.  bool / int8 / uint8 keys are BinnerInteger from the start (vaex/groupby.py:
    593-595): bin_values [False, True, null] resp. arange + null as a masked int64 array (:166-187)"""
    k = np.asarray(values)
    if source_kind == "bool":
        return np.ma.array(k.astype(bool), mask=np.zeros(len(k), dtype=bool), shrink=False)
    if source_kind in ("int8", "uint8"):
        return np.ma.array(k.astype(np.int64), mask=np.zeros(len(k), dtype=bool), shrink=False)
    if len(k) == 0:   # (no group: vaex hands back an empty column of the key's own type)
        return k.astype(source_kind) if source_kind else k
    vmin, vmax = int(k.min()), int(k.max())
    distinct = len(k) if np.all(k[1:] > k[:-1]) or np.all(k[1:] < k[:-1]) else len(np.unique(k))
    if vmax - vmin + 1 <= distinct * 4 / 3:
        return np.ma.array(k.astype(np.int64), mask=np.zeros(len(k), dtype=bool), shrink=False)
    for dt in (np.int8, np.int16, np.int32, np.int64):
        if vmin >= np.iinfo(dt).min and vmax <= np.iinfo(dt).max:
            return k.astype(dt)
    return k


class _NeedsTask(Exception):
    """the call is inside the device groupby's signature, but a column is not one numpy array: answer it as a task of an executor pass"""


class _Plan:
    """what a df.groupby(by, agg) call needs from the device groupby, decided before a row is read"""
    __slots__ = ("by", "agg", "sort", "srt", "asc", "columns", "key_names", "actions", "spec", "selection", "rows", "predicates", "streamed", "key_object", "finishers")


def _plan(df, by, agg, sort=False, ascending=True, row_limit=None, for_task=False):
    """the call's _Plan, or _Decline.  for_task: the rows will come from the executor's chunks (already compacted by the frame's filter:
    the task is pre-filtered like every other task of a filtered frame), so the filter is not planned as a device predicate"""
    import vaex
    import vaex.groupby
    if row_limit is not None:
        raise _Decline("row_limit")
    if by is None:
        raise _Decline("no key")
    by_list = [by] if isinstance(by, str) or not isinstance(by, collections.abc.Iterable) else list(by)
    if not 1 <= len(by_list) <= 8:
        raise _Decline(f"{len(by_list)} keys")
    key_object = None
    for i, b in enumerate(by_list):
        if isinstance(b, vaex.groupby.BinnerBase):
            by_list[i], key_object = _binner_object_key(df, b, len(by_list), run_pending=not for_task)
    asc = list(ascending) if isinstance(ascending, (list, tuple)) else [ascending] * len(by_list)
    srt = list(sort) if isinstance(sort, (list, tuple)) else [sort] * len(by_list)
    if key_object is not None:   # (the object's own order, not the call's: vaex/groupby.py:632-636 passes sort / ascending to the groupers IT makes)
        srt, asc = ([True], [not key_object["invert"]]) if key_object["kind"] == "integer" else ([False], [True])
    if len(set(zip(srt, asc))) > 1:
        raise _Decline("keys sorted in different directions")
    columns, key_names = {}, []
    for b in by_list:
        name, ar = _real_column(df, vaex.utils._ensure_string_from_expression(b), _KEY_KINDS, "group key")
        if name in columns:
            raise _Decline("the same key twice")
        if df.is_category(name) and not (key_object is not None and key_object["kind"] == "grouper"):   # (a Grouper OBJECT groups the codes like any integer column)
            raise _Decline(f"group key {name!r} is categorical")   # (vaex's GrouperCategory hands back the LABELS, and a group per category: vaex/groupby.py:384-442)
        if key_object is not None and key_object["kind"] == "integer" and ar.dtype.name not in _TINY_KEYS:
            raise _Decline("binner object as key (BinnerInteger)")
        if ar.dtype.name in _TINY_KEYS and len(by_list) > 1:
            raise _Decline(f"{ar.dtype.name} key next to other keys")   # (BinnerInteger's N is the dtype's range, not the distinct keys: vaex's combine decision differs)
        columns[name] = ar
        key_names.append(name)
    actions = _normalise_actions(df, key_names, agg)
    if not actions:
        raise _Decline("no aggregation")
    spec, predicates, finishers = {}, {}, {}
    for out_name, aggregate in actions:
        if out_name in spec or out_name in finishers or out_name in key_names:
            raise _Decline("duplicate output column")
        if isinstance(aggregate, _expression_types()):
            # arithmetic over aggregators (vaex.agg.sum('x') / vaex.agg.count(), -vaex.agg.mean('y'), ...: vaex/agg.py:77-189): the leaves are
            # aggregations of the same pass under hidden names, the operators run over their per-group columns when the groups exist
            finishers[out_name] = _translate_tree(df, aggregate, columns, predicates, spec)
        else:
            spec[out_name] = _translate(df, aggregate, columns, predicates)
    # a filtered frame (df[df.x > 0].groupby(...)): vaex compacts every chunk of every column with numpy before its two passes see a row
    # (vaex/execution.py:515-523); here the filter is a device predicate in every aggregator's keep-mask (vaex_amd/vaex_filter.py) and
    # groups without a row inside it are dropped — when it is in the predicate subset over real numeric columns; else vaex's own code
    selection = None
    if df.filtered and not for_task:
        from . import vaex_filter
        pred = vaex_filter.filter_plan(df)
        if pred is None:
            raise _Decline("filtered DataFrame (filter outside the device predicate subset)")
        selection = vaex_filter.filter_expression(df)
        predicates[selection] = pred
        for c in pred.columns:
            if c not in columns:
                columns[c] = _real_column(df, c, tuple(k for k in vaex_filter._NUMERIC if k != "bool"), "filter column")[1]
    plan = _Plan()
    plan.by, plan.agg, plan.sort, plan.srt, plan.asc = by, agg, sort, srt, asc
    plan.columns, plan.key_names, plan.actions, plan.spec, plan.selection = columns, key_names, actions, spec, selection
    plan.rows = len(next(iter(columns.values())))
    plan.predicates = predicates
    plan.streamed = any(isinstance(c, _Streamed) for c in columns.values())
    plan.key_object = key_object
    plan.finishers = finishers
    return plan


def _run(plan, frame):
    """the groups of `frame` (a binned.Frame over the plan's columns): {column: array}, or _Decline"""
    key_names = plan.key_names
    try:
        frame.last_groupby_info = None
        frame._predicates.update(plan.predicates)   # (compiled against the DataFrame: virtual columns are inlined there)
        return frame.groupby(key_names if len(key_names) > 1 else key_names[0], plan.spec, selection=plan.selection)
    except (NotImplementedError, ValueError) as e:
        raise _Decline(str(e))
    except (RuntimeError, MemoryError) as e:
        # a failure the device path reports (HBM exhausted by the device copies or the partition queues, a HIP error): vaex's own two
        # passes answer — chunk by chunk, on the HIP classes where those still work — instead of the call dying here
        drop_device_copies()
        raise _Decline(f"device groupby failed: {type(e).__name__}: {str(e)[:200]}")


def _finish(df, plan, frame, res):
    """the grouped DataFrame the way GroupBy.agg builds it (vaex/groupby.py:955-983)"""
    return _frame_from(df, plan, _finish_arrays(df, plan, frame, res))


def _frame_from(df, plan, finished):
    """finished = {"arrays": {column: array}, "combined": bool} (what the groupby task's result is, and what vaex's task cache keeps) -> the DataFrame"""
    import vaex
    import vaex.dataset
    import vaex.groupby
    dataset_arrays = vaex.dataset.DatasetArrays(dict(finished["arrays"]))
    dataset = vaex.groupby.DatasetGroupby(dataset_arrays, df, plan.by, plan.agg, combine=finished["combined"], expand=True, sort=plan.sort)
    return vaex.from_dataset(dataset)


def _finish_arrays(df, plan, frame, res):
    """the result columns typed and ordered the way GroupBy.agg hands them back, as plain arrays"""
    key_names, columns, actions = plan.key_names, plan.columns, plan.actions
    descending = bool(plan.srt[0]) and not plan.asc[0]
    out = {}
    typed = {name: _key_column_like_vaex(np.asarray(res[name]), columns[name].dtype.name) for name in key_names}
    # several keys: vaex packs them into one grouper when the cartesian product of the key sets is sparsely occupied (< 10 rows
    # per cell, combine='auto': vaex/groupby.py:660-672); the key columns then come back through arrow, as plain arrays
    cells = 1
    for name in key_names:
        cells *= max(1, len(np.unique(np.ma.getdata(typed[name]))))
    combined = len(key_names) >= 2 and plan.rows / cells < 10
    order = None
    if plan.key_object is not None and plan.key_object["kind"] == "grouper":
        # a Grouper object's groups are its bin_values in ITS order (a single Grouper is `dense`: every bin is a row of the result,
        # vaex/groupby.py:949-953) — the device's groups, ascending, are looked up bin by bin; a bin without a row (the object was made over
        # other rows than the frame has now) is vaex's own business
        bins = plan.key_object["bin_values"]
        mine = np.asarray(res[key_names[0]]).astype(np.int64)
        order = np.searchsorted(mine, bins.astype(np.int64))
        if len(bins) != len(mine) or (len(mine) and (order.max() >= len(mine) or not np.array_equal(mine[order], bins.astype(np.int64)))):
            raise _Decline("binner object as key (Grouper whose bins are not the frame's groups)")
        typed[key_names[0]] = bins
    for name in key_names:
        k = np.ma.getdata(typed[name]) if combined else typed[name]
        k = k[::-1] if descending else k
        out[df[name]._label] = k
    for out_name, _ in actions:
        v = _finish_tree(plan.finishers[out_name], res) if out_name in plan.finishers else np.asarray(res[out_name])
        out[out_name] = v[order] if order is not None else (v[::-1] if descending else v)
    last.clear()
    fused = frame.last_groupby_info and not frame.last_groupby_info.get("dense")   # (a dense range with its heavy keys peeled off leaves an info too)
    # (a frame over host columns hands its chunks to whichever of its thread slots is free: the kernel names are those of the slots the last pass used)
    ran = sorted({frame.sa.last_kernel(t) for t in getattr(frame, "last_slots", [0])} - {""}) if hasattr(frame.sa, "last_kernel") else []
    kernel = "gb_scatter+gb_reduce" if fused else "+".join(ran)
    last.update(path="device", kernel=kernel, info=frame.last_groupby_info, groups=len(next(iter(out.values()))))
    return {"arrays": out, "combined": bool(combined)}


def fast_groupby(df, by, agg, sort=False, ascending=True, row_limit=None):
    """the grouped DataFrame, or _Decline"""
    plan = _plan(df, by, agg, sort=sort, ascending=ascending, row_limit=row_limit)
    if plan.streamed:
        raise _NeedsTask()   # (arrow / proxy columns: their chunks come through the executor — the caller schedules the groupby task and runs the pass)
    try:
        frame = _frame_for(df, plan.columns)
    except (NotImplementedError, ValueError) as e:
        raise _Decline(str(e))
    except (RuntimeError, MemoryError) as e:   # (see _run)
        drop_device_copies()
        raise _Decline(f"device groupby failed: {type(e).__name__}: {str(e)[:200]}")
    res = _run(plan, frame)
    return _finish(df, plan, frame, res)


# ---------------------------------------------------------------------------------------------------------------------
# device-resident copies of registered columns: a column handed to vaex_amd.cache_columns() is immutable by contract, so a
# groupby over it may keep its bytes in HBM between calls (the same promise the C-level chunk cache rests on)
# ---------------------------------------------------------------------------------------------------------------------
_device_copies = {}
#: plain host columns smaller than this (bytes, all of a call's columns together) are not worth an upload of their own
upload_min_bytes = 256 << 20


def _frame_for(df, columns):
    from . import _cached_arrays
    cols = {}
    for name, ar in columns.items():
        key = ar.__array_interface__["data"][0]
        base = _cached_arrays.get(key)
        if base is not None and base.nbytes == ar.nbytes:
            hit = _device_copies.get(key)
            if hit is None or hit[0] is not base:
                import torch
                import vaex_amd
                hit = (base, torch.from_numpy(np.ascontiguousarray(base)).to(f"cuda:{int(vaex_amd.superagg.config_get('device'))}"))
                _device_copies[key] = hit
            cols[name] = hit[1]
        else:
            cols[name] = ar
    # Plain (unregistered) host columns: the groupby reads the key column twice (its exact range, then the aggregation) and every
    # pass over a whole pageable column from ONE thread crosses PCIe at a third of what the bus gives.  Round 4: such a call uploads
    # each column once, with several copy threads (vxh_upload), into device memory that lives for this call only, and runs on the
    # device copies — 4e8 rows x {int64 key, float64 value}: 216 -> 151 ms (profiles/r04_vaex_dropin_timing.txt).  (Registered columns keep their copies between calls, above.)
    host = [name for name, c in cols.items() if not binned._is_device(c)]
    if host:
        try:
            import torch
            import vaex_amd
            sa = vaex_amd.superagg
            total = sum(cols[name].nbytes for name in host)
            device = int(sa.config_get("device"))   # (the LIBRARY's device, vxh_set_device: not necessarily torch's current one)
            free, _ = torch.cuda.mem_get_info(device)
            # (no "bool": a bool key is told from a uint8 one by its numpy dtype, so such frames stay on the host path — decided HERE,
            #  before anything crosses PCIe; ADVICE r4: the check used to come after the upload)
            kinds = {"int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8,
                     "float32": torch.float32, "float64": torch.float64}
            plain = all(isinstance(cols[name], np.ndarray) and not np.ma.isMaskedArray(cols[name]) and cols[name].dtype.name in kinds and cols[name].dtype.isnative for name in host)
            if plain and total >= upload_min_bytes and total * 4 < free:
                up = {}
                jobs = []
                for name in host:
                    a = np.ascontiguousarray(cols[name])
                    t = torch.empty(a.shape, dtype=kinds[a.dtype.name], device=f"cuda:{device}")
                    jobs.append((a, t))
                    up[name] = t
                if len(jobs) > 1:   # (the columns cross PCIe side by side: upload() releases the GIL)
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(len(jobs)) as pool:
                        list(pool.map(lambda j: sa.upload(j[0], j[1], 6), jobs))
                else:
                    sa.upload(jobs[0][0], jobs[0][1])
                cols.update(up)
                stats["uploaded"] = stats.get("uploaded", 0) + 1
        except (ImportError, RuntimeError, MemoryError):
            pass   # (no room, no torch: the host columns go through the chunk passes as before)
    if len({binned._is_device(c) for c in cols.values()}) > 1:  # (the fused pass wants keys and values in one place)
        cols = dict(columns)
    from . import vaex_dist
    return binned.Frame(cols, comm=vaex_dist.comm())   # (install(distributed=True): the ranks agree on key ranges / unions and merge their partial groups)


def drop_device_copies():
    _device_copies.clear()


# ---------------------------------------------------------------------------------------------------------------------
# df.groupby(..., delay=True): the device groupby as a TASK of vaex's executor (vaex/tasks.py, vaex/cpu.py, vaex/execution.py:343-470).
# A delayed call asks for its work to share ONE pass over the data with whatever else the caller has scheduled (df.mean(delay=True), other
# groupbys, ...) and to be fulfilled by df.execute().  The task names the plan's columns as its expressions; the executor evaluates and —
# for a filtered frame — compacts them chunk by chunk like it does for every task, hands each chunk to the task part from its pool threads,
# and the part appends the chunk to the plan's columns IN HBM (vxh_upload, outside the lock that hands out row offsets: the order of the
# rows does not matter to a groupby).  When the pass is over, `get_result` runs the same device groupby the eager path runs, on the
# assembled columns.  Progress, cancellation and the pass count are the executor's.
# ---------------------------------------------------------------------------------------------------------------------
_PLANS = {}
_tokens = itertools.count(1)
_TORCH_KINDS = ("int8", "int16", "int32", "int64", "uint8", "float32", "float64")


class DeviceCollector:
    """the plan's columns, assembled in HBM from the chunks the executor hands over"""

    def __init__(self, plan, capacity):
        import torch
        import vaex_amd
        self.sa = vaex_amd.superagg
        for name, ar in plan.columns.items():
            if ar.dtype.name not in _TORCH_KINDS:
                raise _Decline(f"delayed groupby: column {name!r} has dtype {ar.dtype} (no device column of that type)")
        total = sum(ar.dtype.itemsize for ar in plan.columns.values()) * capacity
        device = int(self.sa.config_get("device"))   # (the LIBRARY's device, vxh_set_device: the kernels run there, so the columns live there)
        free, _ = torch.cuda.mem_get_info(device)
        if total * 4 >= free:
            raise _Decline("delayed groupby: the columns do not fit the device next to the partition queues")
        self.capacity = capacity
        self.dtypes = {name: ar.dtype for name, ar in plan.columns.items()}
        # (allocated here, on the scheduling thread, and held until the task has run or is dropped: an allocation from inside the pass — the
        #  pool's threads — ended in a GPU memory fault on the one box it was tried on)
        self.cols = {name: torch.empty(capacity, dtype=getattr(torch, ar.dtype.name), device=f"cuda:{device}") for name, ar in plan.columns.items()}
        self.rows = 0
        self.lock = threading.Lock()

    def append(self, chunks):
        n = len(next(iter(chunks.values())))
        if n == 0:
            return
        with self.lock:
            at = self.rows
            self.rows += n
        if at + n > self.capacity:
            raise RuntimeError("delayed groupby: more rows than the frame has")
        for name, block in chunks.items():
            a = np.ascontiguousarray(np.asarray(block), dtype=self.dtypes[name])
            self.sa.upload(a, self.cols[name][at:at + n], 2)

    def frame(self):
        from . import vaex_dist
        return binned.Frame({name: t[:self.rows] for name, t in self.cols.items()}, comm=vaex_dist.comm())


def _collector_for(plan, capacity):
    return DeviceCollector(plan, capacity)


def _block_as_numpy(block):
    """a chunk as the executor hands it over -> a plain numpy array; missing values are outside the device groupby (RuntimeError: the task
    then leaves the pass to the others and vaex's own groupby answers afterwards)"""
    if isinstance(block, np.ndarray):
        if np.ma.isMaskedArray(block):
            if np.ma.getmaskarray(block).any():
                raise RuntimeError("delayed groupby: a chunk with masked values")
            return np.ma.getdata(block)
        return block
    if hasattr(block, "null_count") and hasattr(block, "type"):   # a pyarrow Array / ChunkedArray (arrow-backed frames)
        if block.null_count:
            raise RuntimeError("delayed groupby: a chunk with missing values")
        import vaex.array_types
        return np.asarray(vaex.array_types.to_numpy(block))
    raise RuntimeError(f"delayed groupby: a chunk of type {type(block).__name__} (the plan saw plain numeric columns)")


def _could_be_served(df, by, row_limit):
    """cheap look at a groupby WITHOUT aggregation: only integer key columns the device groupby takes make the lazy object worth it"""
    import vaex
    import vaex.groupby
    if row_limit is not None or by is None:
        return False
    if df.filtered:
        from . import vaex_filter
        if vaex_filter.filter_plan(df) is None:
            return False
    by_list = [by] if isinstance(by, str) or not isinstance(by, collections.abc.Iterable) else list(by)
    if not 1 <= len(by_list) <= 8 or any(isinstance(b, vaex.groupby.BinnerBase) for b in by_list):
        return False
    try:
        for b in by_list:
            _real_column(df, vaex.utils._ensure_string_from_expression(b), _KEY_KINDS, "group key", materialise=False)   # (a cheap look: nothing is evaluated here)
    except (_Decline, Exception):
        return False
    return True


def _served(progress, fn):
    """fn() under vaex's progress protocol (vaex/progress.py: a bool, a name or a callable f(fraction) whose False cancels): the device
    groupby is ONE uninterruptible pass, so the callback is asked before it starts — a False there is vaex's UserAbort
    (vaex/execution.py:UserAbort) — and told 1.0 when the result exists"""
    if progress is None or progress is False:
        return fn()
    import vaex.execution
    import vaex.utils
    bar = vaex.utils.progressbars(progress, title="groupby")
    if bar(0.0) is False:
        raise vaex.execution.UserAbort("cancelled")
    try:
        result = fn()
    except _Decline:
        raise
    bar(1.0)
    return result


def install(vaex_module, state):
    import vaex.dataframe
    import vaex.groupby
    import vaex.promise
    cls = vaex.dataframe.DataFrameLocal   # (vaex/dataframe.py:7133: groupby is defined on the local frame)
    original = cls.groupby

    def declined(e):
        last.clear()
        last.update(path="vaex", why=str(e))
        stats["vaex"] += 1
        stats["why"][str(e)[:100]] = stats["why"].get(str(e)[:100], 0) + 1
        if os.environ.get("VAEX_AMD_GROUPBY_TRACE"):   # (which caller: the reference's test id when its suite runs under install())
            stats.setdefault("by_test", {}).setdefault(os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], []).append(str(e)[:70])

    import vaex.cpu
    import vaex.tasks

    class TaskGroupbyHip(vaex.tasks.Task):
        """df.groupby(by, agg, delay=True) as one task of the executor's pass (see above); fulfilled with the grouped DataFrame"""
        snake_name = "groupby_hip"
        see_all = True        # ONE task part, shown every chunk (vaex/execution.py:404-406, :553-556)
        # Round 6: like every task of the reference (vaex/execution.py:227-237, :457-474) — with vaex.cache on, a groupby whose frame, keys and
        # aggregations were seen before is fulfilled from the cache when it is scheduled, without a pass.  The task's result is the finished
        # columns as plain arrays (what the cache keeps: any backend can pickle them); the caller's promise turns them into the DataFrame.
        cacheable = True

        def __init__(self, df, plan, token):
            super().__init__(df=df, expressions=list(plan.columns), pre_filter=df.filtered, name=self.snake_name)
            self.selections = []
            self.plan, self.token = plan, token

        def fingerprint(self):
            # (the reference hashes the task's ENCODING, vaex/tasks.py:108-114 — here a per-call token that finds the plan again, so the call
            #  itself is hashed: keys, output columns and their aggregations / selections, the filter, the order)
            if self._fingerprint is None:
                import vaex.cache
                plan = self.plan
                call = [list(plan.key_names), [[name, d.name, d.column, None if d.selection is None else str(d.selection)] for name, d in plan.spec.items()],
                        None if plan.selection is None else str(plan.selection), [bool(x) for x in plan.srt], [bool(x) for x in plan.asc],
                        None if plan.key_object is None else [plan.key_object["kind"], vaex.cache.fingerprint(plan.key_object.get("bin_values"))],
                        [[name, repr(a)] for name, a in plan.actions if name in plan.finishers]]
                df_fp = self.df.fingerprint(dependencies=self.dependencies())
                self._fingerprint = f"task-{self.name}-{vaex.cache.fingerprint(call)}-{df_fp}"
            return self._fingerprint

        def get_bin_count(self):
            return 0

        def encode(self, encoding):
            return {"token": self.token}

        def __repr__(self):
            return f"task-{self.snake_name}: by={self.plan.key_names!r}"

    class TaskPartGroupbyHip(vaex.cpu.TaskPart):
        snake_name = "groupby_hip"

        def __init__(self, df, token):
            plan, collector, fallback, task = _PLANS[token]
            super().__init__(df, list(plan.columns), self.snake_name, df.filtered)
            self.token, self.plan, self.collector, self.fallback, self.task = token, plan, collector, fallback, task
            self.failed = None

        @classmethod
        def decode(cls, encoding, spec, df, nthreads):
            return cls(df, spec["token"])

        def ideal_splits(self, nthreads):
            return 1

        def memory_usage(self):
            return 0

        def process(self, thread_index, i1, i2, filter_mask, selection_masks, blocks):
            # (called from the pool's threads, several at a time: the collector hands out row ranges under its lock)
            if self.failed is not None:
                return
            try:
                self.collector.append({name: _block_as_numpy(b) for name, b in zip(self.plan.columns, blocks)})
            except Exception as e:
                # (HBM exhausted, a HIP error, a chunk the plan did not expect — a TypeError / ValueError from its conversion included: the
                #  pass goes on for the caller's other tasks; this task is answered by vaex's own groupby when the pass is over.  An exception
                #  leaving process() would cancel EVERY task of the pass: vaex/execution.py:567-570)
                self.failed = e

        def reduce(self, others):
            pass

        def get_result(self):
            _PLANS.pop(self.token, None)
            try:
                if self.failed is not None:
                    drop_device_copies()
                    raise _Decline(f"device groupby failed: {type(self.failed).__name__}: {str(self.failed)[:200]}")
                if self.collector.rows == 0:
                    raise _Decline("delayed groupby: the filter left no row")   # (vaex's own answer: no group, its own column types)
                frame = self.collector.frame()
                res = _run(self.plan, frame)
                result = _finish_arrays(self.df, self.plan, frame, res)
            except _Decline as e:
                # the data turned out to be outside the device groupby (key ranges whose product overflows, a device failure, ...): the pass
                # is over and the executor idle (vaex/execution.py:436-441) — vaex's own groupby answers, now
                declined(e)
                task = self.task() if self.task is not None else None
                if task is not None:
                    task.cacheable = False      # (a DataFrame of vaex's own making is not what this task's cache entries are)
                result = {"frame": self.fallback()}
            else:
                stats["task"] += 1
            finally:
                self.collector = None
            return result

    vaex.tasks.register(TaskGroupbyHip)
    vaex.cpu.register(TaskPartGroupbyHip)

    def schedule_task(df, by, agg, sort, ascending, row_limit, kwargs):
        """the scheduled TaskGroupbyHip (a promise of the grouped DataFrame), or _Decline"""
        from . import vaex_dist
        if vaex_dist.active():
            raise _Decline("delayed groupby on a row-sharded frame")
        plan = _plan(df, by, agg, sort=sort, ascending=ascending, row_limit=row_limit, for_task=True)
        if plan.rows == 0:
            raise _Decline("empty frame")
        try:
            collector = _collector_for(plan, plan.rows)
        except (RuntimeError, MemoryError, ImportError) as e:
            raise _Decline(f"device groupby failed: {type(e).__name__}: {str(e)[:200]}")
        token = next(_tokens)
        task = TaskGroupbyHip(df, plan, token)
        _PLANS[token] = (plan, collector, lambda: original(df, by=by, agg=agg, delay=False, **kwargs), weakref.ref(task))
        weakref.finalize(task, _PLANS.pop, token, None)   # (a task that is dropped, cancelled or rejected before its part is built)
        scheduled = df.executor.schedule(task)            # (the task itself, an equal one already waiting, or — vaex.cache on — this one, fulfilled from the cache)
        if scheduled is not task or scheduled.isFulfilled:
            _PLANS.pop(token, None)                       # (no part will be built for this token: its collector's HBM goes now)
            if scheduled.isFulfilled:
                stats["cached"] = stats.get("cached", 0) + 1
        return scheduled.then(lambda finished: finished["frame"] if "frame" in finished else _frame_from(df, plan, finished))

    def eager(df, by, actions, sort, ascending, row_limit, kwargs, progress):
        """the grouped DataFrame of an eager call, or _Decline: one fused pass over whole numpy columns — or, where a column is held in another
        container (arrow, a dataset's proxy: round 6), the same device groupby fed by ONE pass of the executor, the groupby task collecting the
        chunks in HBM (vaex's own groupby takes two passes); what else the caller has scheduled rides that pass, as with vaex's own"""
        try:
            result = _served(progress, lambda: fast_groupby(df, by, actions, sort=sort, ascending=ascending, row_limit=row_limit))
        except _NeedsTask:
            promise = schedule_task(df, by, actions, sort, ascending, row_limit, kwargs)
            before = stats["task"]
            df.execute()
            result = promise.get()
            if stats["task"] == before:   # (the task handed the call back to vaex after the pass — a chunk with missing values, ...: booked as declined there)
                raise _Answered(result)
            stats["task"] -= 1            # (an eager call: the caller books it under "device")
        return result

    class _Answered(Exception):
        """vaex's own groupby answered inside the task (and the decline is booked): nothing left to do but hand the result on"""

        def __init__(self, result):
            self.result = result

    class LazyGroupBy(vaex.groupby.GroupBy):
        """df.groupby(by) WITHOUT agg: vaex builds the groupers — the distinct-key pass over the key columns — in GroupBy.__init__
        (vaex/groupby.py:602-668), before it knows the aggregation.  This object postpones that: `.agg(...)` of a signature the device
        groupby takes is answered by it (one fused pass, no groupers at all); anything else — another method, an attribute, an
        aggregation outside the signature — first becomes the real GroupBy (same arguments, vaex's own constructor) and carries on as
        that."""

        def __init__(self, df, kwargs):
            self.__dict__["_lazy"] = (df, kwargs)

        def _materialise(self):
            df, kw = self.__dict__.pop("_lazy")
            real = original(df, agg=None, delay=False, **kw)
            self.__dict__.update(real.__dict__)

        def __getattr__(self, name):   # (only reached when the attribute is not there: everything GroupByBase.__init__ sets)
            if "_lazy" in self.__dict__ and not (name.startswith("__") and name.endswith("__")):
                self._materialise()
                return getattr(self, name)
            raise AttributeError(name)

        def agg(self, actions, delay=False, progress=None):
            if "_lazy" in self.__dict__ and delay:
                # (delay=True is a request to batch this aggregation with the caller's other tasks into one pass of the executor —
                #  vaex/groupby.py:975-1017: the device groupby joins that pass as a task)
                df, kw = self.__dict__["_lazy"]
                try:
                    return schedule_task(df, kw["by"], actions, kw["sort"], kw["ascending"], kw["row_limit"],
                                         dict(sort=kw["sort"], ascending=kw["ascending"], assume_sparse=kw["assume_sparse"], row_limit=kw["row_limit"], copy=kw["copy"], progress=progress if progress is not None else kw.get("progress")))
                except _Decline as e:
                    declined(e)
                    self._materialise()
            if "_lazy" in self.__dict__:
                df, kw = self.__dict__["_lazy"]
                try:
                    result = eager(df, kw["by"], actions, kw["sort"], kw["ascending"], kw["row_limit"],
                                   dict(sort=kw["sort"], ascending=kw["ascending"], assume_sparse=kw["assume_sparse"], row_limit=kw["row_limit"], copy=kw["copy"], progress=progress if progress is not None else kw.get("progress")),
                                   progress if progress is not None else kw.get("progress"))
                except _Answered as a:
                    return df._delay(delay, vaex.promise.Promise.fulfilled(a.result))
                except _Decline as e:
                    declined(e)
                    self._materialise()
                else:
                    stats["device"] += 1
                    return df._delay(delay, vaex.promise.Promise.fulfilled(result))
            return vaex.groupby.GroupBy.agg(self, actions, delay=delay, progress=progress)

    def groupby(self, by=None, agg=None, sort=False, ascending=True, assume_sparse="auto", row_limit=None, copy=True, progress=None, delay=False):
        if agg is not None and delay:
            # (a delayed groupby shares the pass of df.execute() with the caller's other tasks: the device groupby as a task of that pass —
            #  or, outside its signature, vaex's own delayed tasks on the HIP classes; the reference's tests count the passes:
            #  tests/groupby_test.py:598-606)
            try:
                return schedule_task(self, by, agg, sort, ascending, row_limit,
                                     dict(sort=sort, ascending=ascending, assume_sparse=assume_sparse, row_limit=row_limit, copy=copy, progress=progress))
            except _Decline as e:
                declined(e)
        elif agg is not None:
            try:
                result = eager(self, by, agg, sort, ascending, row_limit,
                               dict(sort=sort, ascending=ascending, assume_sparse=assume_sparse, row_limit=row_limit, copy=copy, progress=progress), progress)
            except _Answered as a:
                return self._delay(delay, vaex.promise.Promise.fulfilled(a.result))
            except _Decline as e:
                declined(e)
            else:
                stats["device"] += 1
                return self._delay(delay, vaex.promise.Promise.fulfilled(result))
        elif not delay and _could_be_served(self, by, row_limit):
            return LazyGroupBy(self, dict(by=by, sort=sort, ascending=ascending, assume_sparse=assume_sparse, row_limit=row_limit, copy=copy, progress=progress))
        return original(self, by=by, agg=agg, sort=sort, ascending=ascending, assume_sparse=assume_sparse, row_limit=row_limit, copy=copy, progress=progress, delay=delay)

    groupby.__doc__ = original.__doc__
    groupby.__wrapped__ = original
    cls.groupby = groupby
    state["groupby"] = (cls, original)
    state["groupby_task"] = (TaskGroupbyHip, TaskPartGroupbyHip)


def uninstall(vaex_module, state):
    cls, original = state["groupby"]
    cls.groupby = original
    drop_device_copies()
