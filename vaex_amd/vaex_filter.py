"""Filtered frames of an unmodified vaex without the per-chunk compaction ("includes filter compaction").

vaex's aggregation tasks are pre-filtered (vaex/tasks.py:508 `pre_filter=df.filtered`): for `dff = df[df.x > 0]` the executor
evaluates the filter per chunk and then copies EVERY column the tasks read through a boolean index with numpy
(vaex/execution.py:515-523, `vaex.array_types.filter`) before a task part sees a row.  Behind `install()` that is the wrong way
round twice: the compaction costs more host time than the aggregation costs device time, and it hands the task part fresh
temporaries — the device column cache (keyed by host address, vaex_amd.cache_columns) can never hit for a filtered frame.

A binned aggregation does not need compacted rows; a keep-mask says the same.  So:

  * `vaex.execution.Run.__init__` is wrapped: when EVERY task a run holds for a filtered frame is a `TaskAggregations` whose
    aggregators read a keep-mask the way their unfiltered form does (count / sum / sum-moment / min / max: not AggFirst / AggList /
    AggNUnique, whose masks the reference indexes block-locally, src/agg_first.cpp:131), those tasks run with pre_filter = False:
    the executor then evaluates the filter exactly as before (cached per chunk in df._selection_mask_caches) but hands the task part
    the UNCOMPACTED blocks plus the full-length filter mask (vaex/execution.py:535-536, :573).  A run that mixes such tasks with
    others (df.minmax, the distinct-key pass of a groupby, ...) is left alone — vaex refuses mixed runs (:62-66) — and so is every run of a
    frame that carries functions of its own (add_function / apply): such a function may rely on never seeing a filtered-out row;
  * the registered task part (TaskPartAggregationHip) makes the filter part of every aggregator's keep-mask: as a DEVICE predicate
    when the filter is in the comparison subset of vaex_amd.predicate (alone, or `(filter) & (selection)` in one predicate when the
    aggregation's selection is on the device too and both fit its four terms) — then no mask byte crosses PCIe — otherwise by AND-ing
    the executor's host mask into the selection masks it passes to TaskPartAggregation.process (vaex/cpu.py:735-784);
  * a task part that ended up on vaex's own C++ classes does the same with host masks: nothing is lost, only not accelerated.

Results are those of the pre-filtered path: the same rows reach the same cells (tests/test_vaex_filter.py, against vaex's C++ with
pre_filter=True in the same process)."""
import numpy as np

from . import predicate as _predicate

#: counters for tests: runs switched to the keep-mask form / left pre-filtered because of other tasks, chunks whose filter was a
#: device predicate / a host mask
stats = {"runs_switched": 0, "runs_mixed": 0, "device_chunks": 0, "host_chunks": 0}

_KEEP_MASK_AGGS = ("AggCount", "AggSum", "AggSumMoment", "AggMin", "AggMax")
_NUMERIC = ("float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool")
_FILTER = "__filter__"   # vaex.dataframe.FILTER_SELECTION_NAME
SPEC_KEY = "hip-filter-as-mask"


def filter_expression(df):
    """the frame's filter as ONE boolean expression string, or None (no filter / not a chain of `&`-combined expressions)"""
    if not df.filtered:
        return None
    sel = df.get_selection(_FILTER)
    parts = []
    while sel is not None:
        if type(sel).__name__ != "SelectionExpression":
            return None
        parts.append(sel.boolean_expression)
        prev = sel.previous_selection
        if prev is not None and sel.mode != "and":   # df[a][b] chains with mode "and" (vaex/dataframe.py:5377); anything else: host mask
            return None
        sel = prev
    return " & ".join(f"({p})" for p in reversed(parts))


def _known_columns(df):
    return {name: ar for name, ar in df.columns.items() if _predicate.plain_numeric_dtype(ar) is not None}


def _virtual_columns(df):
    from . import vaex_selection
    return vaex_selection._virtual_columns(df)


def filter_plan(df):
    """the Predicate the frame's filter compiles to, or None (then the executor's host mask is used)"""
    expr = filter_expression(df)
    if expr is None:
        return None
    try:
        return _predicate.compile_selection(expr, _known_columns(df), virtual=_virtual_columns(df))
    except _predicate.Unsupported:
        return None


def combined_plan(df, selection):
    """`(filter) & (selection)` as one Predicate, or None when it does not fit"""
    expr = filter_expression(df)
    if expr is None:
        return None
    try:
        return _predicate.compile_selection(f"({expr}) & ({selection})", _known_columns(df), virtual=_virtual_columns(df))
    except _predicate.Unsupported:
        return None


def _qualifies(task):
    if type(task).__name__ != "TaskAggregations":
        return False
    return all(getattr(d, "name", None) in _KEEP_MASK_AGGS for d in task.aggregation_descriptions) and len(task.aggregation_descriptions) > 0


def install(vaex_module, state):
    import vaex.execution
    import vaex.tasks
    run_cls = vaex.execution.Run
    run_init = run_cls.__init__

    def __init__(self, tasks):
        per_df = {}
        for task in tasks:
            per_df.setdefault(task.df, []).append(task)
        for df, ts in per_df.items():
            if not df.filtered:
                continue
            # (a frame with functions of its own — df.add_function, apply: the filter may be what keeps rows such a function cannot take away
            #  from it, and only compacted blocks do that: tests/agg_test.py:405-416 `assert 4 not in x` — stays with vaex's compaction)
            if all(_qualifies(t) for t in ts) and not df.functions:
                for t in ts:
                    t.pre_filter = False
                    t.__dict__["_hip_filter_as_mask"] = True
                stats["runs_switched"] += 1
            else:
                for t in ts:   # (a task switched by an earlier, failed run keeps vaex's form here)
                    if t.__dict__.pop("_hip_filter_as_mask", False):
                        t.pre_filter = True
                stats["runs_mixed"] += 1
        run_init(self, tasks)

    run_cls.__init__ = __init__
    task_cls = vaex.tasks.TaskAggregations
    task_encode = task_cls.encode

    def encode(self, encoding):
        spec = task_encode(self, encoding)
        if self.__dict__.get("_hip_filter_as_mask"):
            spec[SPEC_KEY] = True
        return spec

    task_cls.encode = encode
    state["filter"] = (run_cls, run_init, task_cls, task_encode)


def uninstall(vaex_module, state):
    run_cls, run_init, task_cls, task_encode = state["filter"]
    run_cls.__init__ = run_init
    task_cls.encode = task_encode


def mark(part):
    """called once by the task part's decode (after the device selections are attached): every aggregation whose filter is not part of a
    device Selection is a SELECTED aggregation from here on — an unselected one gets the marker `__filter__` as its selection, so that
    TaskPartAggregation.process asks for a mask (vaex/cpu.py:735-745).  Done here and not per chunk: vaex's pool threads run `process`
    of ONE task part concurrently, each on its own thread slot (vaex/execution.py:404-406, :553-556 — `see_all` parts), so the part must
    not change while chunks are in flight.  get_result / reduce only count the entries."""
    on_device = part._hip_filter_on_device
    for i, (desc, selections, aggs, waslist) in enumerate(part.aggregations):
        if i not in on_device:
            part.aggregations[i] = (desc, [_FILTER if (s is None or s is False) else s for s in selections], aggs, waslist)


def process(part, base, thread_index, i1, i2, filter_mask, selection_masks, blocks):
    """TaskPartAggregation.process for uncompacted blocks: the filter joins every aggregator's keep-mask (re-entrant: see mark)"""
    fm = filter_mask
    if np.ma.isMaskedArray(fm):
        import vaex.utils
        fm = vaex.utils.unmask_selection_mask(fm)
    fm = np.asarray(fm)
    if fm.dtype != np.bool_:
        fm = fm.astype(np.bool_)
    on_device = part._hip_filter_on_device
    masks = list(selection_masks)
    g = 0
    host = False
    for i, (desc, selections, aggs, waslist) in enumerate(part.aggregations):
        for s in selections:
            if i in on_device:   # (its device Selection holds the filter already)
                pass
            elif s is _FILTER or (isinstance(s, str) and s == _FILTER):
                masks[g] = fm
                host = True
            else:
                m = masks[g]
                if np.ma.isMaskedArray(m):
                    import vaex.utils
                    m = vaex.utils.unmask_selection_mask(m)
                masks[g] = fm & np.asarray(m).astype(np.bool_, copy=False)
                host = True
            g += 1
    stats["host_chunks" if host else "device_chunks"] += 1
    return base.process(part, thread_index, i1, i2, None, masks, blocks)
