"""Selections of an unmodified vaex evaluated on the device (SURVEY §8 f.2 through the reference's own API).

vaex evaluates `df.count(..., selection="(x > 0) & (v < 3.5)")` per chunk with numpy on the pool threads
(vaex/execution.py:530-549 `selections = [... selection_scope.evaluate(s) for s in task.selections]`, vaex/scopes.py:138-177)
and hands TaskPartAggregation.process one boolean array per aggregation, which becomes the aggregator's keep-mask
(vaex/cpu.py:735-784).  With vaex_amd.install() the comparison subset of vaex_amd.predicate (<= 4 terms `column <op> number`
joined by & | ~) never becomes a host array:

  * TaskAggregations.add_aggregation_operation (vaex/tasks.py:516-541) is wrapped: when an aggregation's selection is such an
    expression over real numeric columns, its entry in `task.selections` becomes None — the executor then evaluates nothing
    for it — and the predicate's columns join the TAIL of `task.expressions_all`, so that the executor loads their chunks
    like any other column the task reads and hands them to the task part behind the blocks it already expects;
  * the registered task part (TaskPartAggregationHip, vaex_amd/__init__.py) finds the same predicates from the encoded
    aggregations, attaches ONE device Selection per distinct predicate to the aggregators concerned (vxh_agg_set_selection:
    rows are kept where the predicate holds AND the data mask, if any, is set) and registers the predicate columns' chunks
    per thread slot in `process`;
  * a task that ends up on vaex's own C++ (an aggregator the HIP classes do not offer) evaluates the same predicate with numpy
    in `process` and passes the mask on — nothing is lost, only not accelerated.

NAMED selections (df.select("x > 0"); df.count(selection=True) — vaex/dataframe.py:5041, the history of SelectionExpression /
SelectionInvert objects per name, vaex/selections.py:40-160) are resolved to ONE boolean expression when the aggregation is
scheduled — `replace` drops the history, `and` / `or` / `subtract` combine with the previous one, select_inverse negates — and
take the same road when that expression is in the subset.  vaex looks a name up when the run starts; so does this: delayed
aggregations enter their TaskAggregations in the executor's merge at the start of the run (vaex/execution.py:_merge ->
add_aggregation_operation), which is where the name is resolved.  The resolved expression travels with the task (SPEC_KEY in
its encoding) and the task part checks it against the frame once more when it is built — should the name mean something else by
then, it raises instead of aggregating rows that are neither definition's.

Selections with missing-value columns, lasso / dropna selections, `xor` histories and expressions outside the subset keep vaex's
host evaluation untouched."""
import numpy as np

from . import predicate as _predicate

#: counters for tests: predicates planned at task-build time, chunks whose predicate ran on the device / on the host fallback
stats = {"planned": 0, "device_chunks": 0, "host_chunks": 0}

_WITH_DEVICE_SELECTION = ("AggCount", "AggSum", "AggSumMoment", "AggMin", "AggMax")   # (vxh_agg_set_selection)
_NUMERIC = ("float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool")


SPEC_KEY = "hip-named-selections"   # {str(aggregation index): the expression its named selection stood for when scheduled}


def _selection_as_expression(sel):
    """a selection object's history as one boolean expression string, or None (vaex/selections.py:_select_functions)"""
    if sel is None:
        return None
    kind = type(sel).__name__
    if kind == "SelectionInvert":
        prev = _selection_as_expression(sel.previous_selection)
        return None if prev is None else f"~({prev})"
    if kind != "SelectionExpression":
        return None
    cur = f"({sel.boolean_expression})"
    if sel.previous_selection is None:   # (mode "replace" drops it in Selection.__init__; the first selection of a name has none)
        return cur
    prev = _selection_as_expression(sel.previous_selection)
    if prev is None:
        return None
    if sel.mode == "and":
        return f"({prev}) & {cur}"
    if sel.mode == "or":
        return f"({prev}) | {cur}"
    if sel.mode == "subtract":
        return f"({prev}) & ~{cur}"
    return None


def named_expression(df, name):
    """the boolean expression the named selection `name` stands for right now, or None"""
    if not df.has_selection(name):
        return None
    return _selection_as_expression(df.get_selection(name))


_known_cache = {}   # id(df.columns dict) -> (fingerprint, names); one entry per live frame, a handful of frames per process


def _known_columns(df):
    """the frame's real numeric columns without missing values (numpy, or arrow without nulls): what a device predicate may compare.
    Remembered per frame while its column set stays the same objects (every scheduled aggregation — and every task part's
    decode — asked again, a scan of all columns and of the arrow null counts each time: O(aggregations^2 x columns) on wide frames)"""
    cols = df.columns
    # (ids alone can be reused after a collection by a column of another type under the same name — ADVICE r4: a stale "plain numeric"
    #  verdict would let a predicate compile over strings or nulls — so the fingerprint carries each column's type and dtype as well)
    fp = tuple((name, id(ar), type(ar).__name__, str(getattr(ar, "dtype", getattr(ar, "type", "")))) for name, ar in cols.items())
    hit = _known_cache.get(id(cols))
    if hit is not None and hit[0] == fp:
        return {name: cols[name] for name in hit[1]}   # (names only are remembered: no array is kept alive by the cache)
    known = {name: ar for name, ar in cols.items() if _predicate.plain_numeric_dtype(ar) is not None}
    if len(_known_cache) > 64:
        _known_cache.clear()
    _known_cache[id(cols)] = (fp, tuple(known))
    return known


def _virtual_columns(df):
    """name -> expression string of the frame's virtual columns (vaex/dataframe.py `virtual_columns`): a selection over one is compiled
    with the expression inlined, when that is arithmetic over float64 columns (vaex_amd.predicate)"""
    try:
        return {str(k): str(v) for k, v in dict(df.virtual_columns).items()}
    except Exception:
        return {}


def plan_for(df, descriptor, resolved=None):
    """the Predicate an aggregation's selection compiles to, or None (then vaex evaluates the selection itself).
    resolved: for a NAMED selection the expression it stood for when the aggregation was scheduled ("" = it was not planned then);
    None = resolve the name now"""
    sel = getattr(descriptor, "selection", None)
    if sel is None or sel is False or isinstance(sel, (list, tuple)):
        return None
    if getattr(descriptor, "name", None) not in _WITH_DEVICE_SELECTION:   # AggFirst / AggList / AggNUnique read host masks only
        return None
    sel = "default" if sel is True else str(sel)
    if df.has_selection(sel):   # a named selection: its history as one expression
        sel = named_expression(df, sel) if resolved is None else resolved
        if not sel:
            return None
    try:
        pred = _predicate.compile_selection(sel, _known_columns(df), virtual=_virtual_columns(df))
    except _predicate.Unsupported:
        return None
    return pred


def _is_named(df, descriptor):
    sel = getattr(descriptor, "selection", None)
    if sel is None or sel is False or isinstance(sel, (list, tuple)):
        return False
    return df.has_selection("default" if sel is True else str(sel))


def _filter_columns(df, descriptors):
    """columns of the frame's filter when it compiles to a device predicate and the task could run with the filter as a keep-mask
    (vaex_amd/vaex_filter.py): they ride behind the selection predicates' columns"""
    from . import vaex_filter
    if not df.filtered or not all(getattr(d, "name", None) in vaex_filter._KEEP_MASK_AGGS for d in descriptors):
        return []
    pred = vaex_filter.filter_plan(df)
    return list(pred.columns) if pred is not None else []


def extras_of(df, descriptors, named):
    """[(aggregation index, Predicate)], and the predicate columns in first-seen order, the filter's last (= the tail of
    task.expressions_all).  named: {str(index): expression} of the aggregations whose NAMED selection was planned when they were
    scheduled — a named selection outside it is never planned here, whatever the name means by now"""
    plans, extras = [], []
    for i, d in enumerate(descriptors):
        p = plan_for(df, d, named.get(str(i), "") if _is_named(df, d) else None)
        if p is not None:
            plans.append((i, p))
            for c in p.columns:
                if c not in extras:
                    extras.append(c)
    for c in _filter_columns(df, descriptors):
        if c not in extras:
            extras.append(c)
    return plans, extras


def install(vaex_module, state):
    import vaex.tasks
    cls = vaex.tasks.TaskAggregations
    original = cls.add_aggregation_operation

    def add_aggregation_operation(self, aggregator_descriptor):
        tail = self.__dict__.get("_hip_extras", [])
        if tail:   # (they sit at the tail: take them off while vaex appends this aggregation's own expressions)
            del self.expressions_all[len(self.expressions_all) - len(tail):]
        task = original(self, aggregator_descriptor)
        named = self.__dict__.setdefault("_hip_named", {})
        pred = plan_for(self.df, aggregator_descriptor)
        if pred is not None:
            self.selections[-1] = None   # the executor evaluates nothing for this aggregation (vaex/execution.py:549)
            stats["planned"] += 1
            if _is_named(self.df, aggregator_descriptor):
                named[str(len(self.aggregation_descriptions) - 1)] = pred.expression
        _, tail = extras_of(self.df, self.aggregation_descriptions, named)   # (the same call the task part's decode makes)
        self.__dict__["_hip_extras"] = tail
        if tail:
            self.expressions_all.extend(tail)
            self.dtypes = {expr: self.df.data_type(expr).index_type for expr in self.expressions_all}
        return task

    cls.add_aggregation_operation = add_aggregation_operation
    original_encode = cls.encode

    def encode(self, encoding):
        spec = original_encode(self, encoding)
        if self.__dict__.get("_hip_named"):
            spec[SPEC_KEY] = dict(self.__dict__["_hip_named"])
        return spec

    cls.encode = encode
    state["selection"] = (cls, original, original_encode)


def uninstall(vaex_module, state):
    cls, original, original_encode = state["selection"]
    cls.add_aggregation_operation = original
    cls.encode = original_encode


def attach(part, backend_used, superagg, nthreads, filter_as_mask=False, named=None):
    """called by the task part's decode: remember the predicates, and (HIP classes) hand them to the aggregators.
    filter_as_mask: the run hands this part uncompacted blocks of a filtered frame (vaex_amd/vaex_filter.py) — where the filter is a
    device predicate it joins the aggregators' Selection here (alone or AND-ed with the aggregation's own predicate).
    named: the task's SPEC_KEY entry — what its named selections stood for when they were scheduled"""
    from . import vaex_filter
    named = dict(named or {})
    for i, expr in named.items():
        name = part.aggregation_descriptions[int(i)].selection
        name = "default" if name is True else str(name)
        if named_expression(part.df, name) != expr:
            raise RuntimeError(f"selection {name!r} was re-defined after an aggregation over it was scheduled (it stood for {expr!r}): "
                               "vaex_amd planned that definition for the device; execute delayed aggregations before changing a selection they use")
    plans, extras = extras_of(part.df, part.aggregation_descriptions, named)
    part._hip_plans, part._hip_extras, part._hip_selections = plans, extras, []
    part._hip_refs = {}   # created ONCE, here: pool threads call process() of one part concurrently and must never rebind it
    part._hip_filter_as_mask = bool(filter_as_mask)
    part._hip_filter_on_device = set()
    fpred = vaex_filter.filter_plan(part.df) if (filter_as_mask and backend_used == "hip") else None
    if plans or fpred is not None:
        _attach_predicates(part, backend_used, superagg, nthreads, plans, extras, fpred)
    if filter_as_mask:
        vaex_filter.mark(part)


def _attach_predicates(part, backend_used, superagg, nthreads, plans, extras, fpred):
    from . import vaex_filter
    objs = {}

    def selection_object(pred):
        key = pred.key()
        if key not in objs:
            dtypes = [_predicate.dtype_code(_predicate.plain_numeric_dtype(part.df.columns[c])) for c in pred.columns]
            objs[key] = superagg.Selection(nthreads, dtypes, [(c, op, v) for c, op, v in pred.terms], pred.truth)
            if pred.programs:   # arithmetic / virtual-column terms (vxh_selection_set_program)
                objs[key].set_programs({t: [tuple(st) for st in steps] for t, steps in pred.programs.items()})
        return objs[key]
    # global index of an aggregation's (single) selection in the executor's `selections` list: one entry per aggregation
    # without a list of selections (vaex/tasks.py:528-535); lists of selections never get a plan
    position = 0
    positions = []
    for desc, selections, aggs, waslist in part.aggregations:
        positions.append(position)
        position += len(selections)
    planned = dict(plans)
    for i, (desc, selections, aggs, waslist) in enumerate(part.aggregations):
        pred = planned.get(i)
        if pred is None and not (fpred is not None and len(selections) == 1 and (selections[0] is None or selections[0] is False)):
            continue   # (a host-evaluated selection / a list of selections: the filter, if any, joins as a host mask in process)
        entry = dict(index=positions[i], pred=pred, sel=None)
        if backend_used == "hip":
            if fpred is not None:
                both = fpred if pred is None else vaex_filter.combined_plan(part.df, pred.expression)
                if both is not None and all(c in extras for c in both.columns):
                    pred = entry["pred"] = both
                    part._hip_filter_on_device.add(i)
            entry["sel"] = selection_object(pred)
            for a in aggs:
                a.set_selection(entry["sel"])
            # the aggregation is "unselected" for the base class from here on: no host mask is expected for it
            part.aggregations[i] = (desc, [None] * len(selections), aggs, waslist)
        part._hip_selections.append(entry)


def _as_numpy(block):
    """a predicate column's chunk as numpy: numpy chunks as they are, arrow chunks (no nulls: plain_numeric_dtype) through vaex's own
    converter — a view of the arrow buffer for the primitive types, an unpacked copy for bool"""
    if isinstance(block, np.ndarray):
        return block
    try:
        import vaex.array_types
        return np.asarray(vaex.array_types.to_numpy(block))
    except ImportError:
        return np.asarray(block)


def before_process(part, thread_index, selection_masks, blocks):
    """-> (selection_masks, blocks) for the base class's process: the predicate columns' chunks go to the device selections
    (or, on vaex's own C++, become numpy masks); the blocks lose their tail"""
    extras = part._hip_extras
    if not extras:
        return selection_masks, blocks
    nbase = len(blocks) - len(extras)
    tail = dict(zip(extras, blocks[nbase:]))
    seen = set()
    for entry in part._hip_selections:
        pred, idx = entry["pred"], entry["index"]
        if selection_masks[idx] is not None and entry["sel"] is not None:
            # the task and this part disagree about the plan (the executor evaluated a selection the part handed to the device): the
            # base class ignores host masks of this aggregation by now and the device Selection would read stale slot pointers
            raise RuntimeError("vaex_amd: a host mask arrived for an aggregation whose selection was attached to the device")
        if selection_masks[idx] is not None:   # (vaex's own C++: the executor evaluated it after all, use its mask)
            continue
        if entry["sel"] is not None:
            if id(entry["sel"]) not in seen:
                seen.add(id(entry["sel"]))
                for ci, c in enumerate(pred.columns):
                    d = np.ascontiguousarray(_as_numpy(tail[c]))
                    entry["sel"].set_data(thread_index, ci, d.view("u1") if d.dtype == np.bool_ else d)
                    part._hip_refs[(thread_index, id(entry["sel"]), ci)] = d   # (borrowed until the slot's next chunk; distinct keys per thread)
                stats["device_chunks"] += 1
        else:
            selection_masks = list(selection_masks)
            selection_masks[idx] = pred.numpy_mask({c: _as_numpy(tail[c]) for c in pred.columns})
            stats["host_chunks"] += 1
    return selection_masks, blocks[:nbase]
