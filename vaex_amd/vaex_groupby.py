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
_FLOAT_KEYS = ("float64", "float32")   # round 6: grouped by their bit patterns (_coded_key)
_TINY_KEYS = ("bool", "int8", "uint8")   # vaex bins these with BinnerInteger straight away (vaex/groupby.py:593-595): no combined grouper, own key typing
_VALUE_KINDS = ("float64", "float32", "int64", "int32", "int16", "int8", "uint32", "uint16", "uint8", "bool")
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
    if hasattr(ar, "null_count") and hasattr(ar, "type"):   # an arrow dictionary column (a categorical key: its indices are the codes)
        import pyarrow as pa
        if pa.types.is_dictionary(ar.type) and not ar.null_count and pa.types.is_integer(ar.type.index_type):
            return np.dtype(ar.type.index_type.to_pandas_dtype())
        return None
    if type(ar).__name__ == "ColumnProxy" and hasattr(ar, "ds"):
        try:
            dt = df.data_type(name)
            if dt.is_encoded:                       # (a categorical column of a _future() frame shows as a dictionary: the chunks are indices into the labels)
                dt = dt.index_type
            if dt.is_numeric or dt == bool:         # (missing values show when the chunks arrive: the task then leaves the pass to vaex)
                dt = np.dtype(dt.numpy)
                return dt if dt.isnative else None
        except Exception:   # noqa: BLE001  (a column type vaex itself cannot describe as numpy)
            return None
    return None


class _Nullable:
    """a numeric / bool column with missing values, split on the host (round 6): `data` (a missing entry holds whatever the container held) and `mask`"""

    def __init__(self, data, mask):
        self.data, self.mask, self.dtype = data, mask, data.dtype

    def __len__(self):
        return len(self.data)


def _split_nullable(ar, i1=0, i2=None):
    """a masked numpy array / a pyarrow Array or ChunkedArray of a primitive numeric or bool type WITH nulls -> _Nullable, else None"""
    if isinstance(ar, np.ndarray):
        if not np.ma.isMaskedArray(ar) or ar.ndim != 1 or not ar.dtype.isnative or ar.dtype.kind not in "biuf":
            return None
        i2 = len(ar) if i2 is None else i2
        return _Nullable(np.ascontiguousarray(np.ma.getdata(ar)[i1:i2]), np.ascontiguousarray(np.ma.getmaskarray(ar)[i1:i2]))
    if getattr(ar, "type", None) is None or not hasattr(ar, "null_count"):
        return None
    import pyarrow as pa
    if not isinstance(ar, (pa.Array, pa.ChunkedArray)) or not (pa.types.is_integer(ar.type) or pa.types.is_floating(ar.type) or pa.types.is_boolean(ar.type)):
        return None
    i2 = len(ar) if i2 is None else i2
    ar = ar.slice(i1, i2 - i1)
    if isinstance(ar, pa.ChunkedArray):
        ar = ar.combine_chunks() if ar.num_chunks else pa.array([], type=ar.type)
        ar = pa.concat_arrays(ar.chunks) if isinstance(ar, pa.ChunkedArray) else ar
    mask = np.asarray(ar.is_null().to_numpy(zero_copy_only=False))
    data = np.asarray(ar.fill_null(False if pa.types.is_boolean(ar.type) else 0).to_numpy(zero_copy_only=False))
    return _Nullable(np.ascontiguousarray(data), np.ascontiguousarray(mask))


def _real_column(df, expression, kinds, what, materialise=True, nullable=False):
    """the numpy array behind `expression` when it names a real, unmasked column of one of `kinds` (active range applied) — or a _Streamed
    stand-in for a real numeric column held in another container (its rows then reach the device through the executor's chunks)"""
    name = str(expression)
    if name not in df.columns:
        return _virtual_column(df, name, kinds, what, materialise, nullable)
    ar = df.columns[name]
    i1, i2 = df._index_start, df._index_end
    if np.ma.isMaskedArray(ar) or not isinstance(ar, np.ndarray):
        kind = "masked numpy" if np.ma.isMaskedArray(ar) else f"{type(ar).__module__.split('.')[0]}.{type(ar).__name__}" + (f"[{ar.type}]" if hasattr(ar, "type") and hasattr(ar, "null_count") else "")
        dt = None if np.ma.isMaskedArray(ar) else _streamed_dtype(df, name, ar)
        if dt is None and nullable and len(ar) <= materialise_max_rows:
            # round 6: missing values (a numpy mask / arrow nulls) — the column is split on the host into data and mask; a key's missing rows
            # become a code of their own (_nullable_key), a value column's become NaN (_nullable_values)
            split = _split_nullable(ar, i1, i2)
            if split is not None:
                if split.dtype.name not in kinds:
                    raise _Decline(f"{what} {name!r} has dtype {split.dtype}")
                return name, (split if split.mask.any() else split.data)
        if dt is None:
            raise _Decline(f"{what} {name!r} is not a plain numpy column ({kind})")
        if dt.name not in kinds:
            raise _Decline(f"{what} {name!r} has dtype {dt}")
        n = len(ar)
        return name, _Streamed(dt, (n if i2 is None else i2) - i1, ar)
    if ar.dtype.name not in kinds or ar.ndim != 1 or (not ar.dtype.isnative and (not nullable or len(ar) > materialise_max_rows)):
        raise _Decline(f"{what} {name!r} has dtype {ar.dtype}")
    if i2 is None:
        i2 = len(ar)
    if i1 != 0 or i2 != len(ar):
        ar = ar[i1:i2]
    if not ar.dtype.isnative:   # (round 6: a byte-swapped column — FITS / big-endian HDF5 — is converted once on the host)
        ar = ar.astype(ar.dtype.newbyteorder("="))
    return name, ar


#: a virtual column / expression is materialised on the host for the device groupby up to this many rows (one numpy array of the frame's length)
materialise_max_rows = 1 << 27


def _virtual_column(df, name, kinds, what, materialise=True, nullable=False):
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
        return name, _real_column(df, target, kinds, what, nullable=nullable)[1]
    if not materialise or len(df) > materialise_max_rows or df.length_original() > materialise_max_rows:
        raise _Decline(f"{what} {label!r} is not a real column")
    try:
        ar = df.evaluate(name, filtered=False, parallel=False)
    except Exception as e:   # noqa: BLE001  (an expression vaex cannot evaluate to an array: its own groupby says so)
        raise _Decline(f"{what} {label!r} is not a real column ({type(e).__name__})")
    if nullable and (np.ma.isMaskedArray(ar) or not isinstance(ar, np.ndarray)):
        split = _split_nullable(ar)
        if split is None and not np.ma.isMaskedArray(ar):   # (an arrow result without nulls)
            dt = __import__("vaex_amd").predicate.plain_numeric_dtype(ar)
            if dt is not None:
                import vaex.array_types
                ar = np.asarray(vaex.array_types.to_numpy(ar))
        if split is not None:
            if not split.mask.any():
                ar = split.data
            elif split.dtype.name in kinds:
                return name, split
            else:
                raise _Decline(f"{what} {label!r} has dtype {split.dtype}")
    if np.ma.isMaskedArray(ar) or not isinstance(ar, np.ndarray) or ar.ndim != 1:
        raise _Decline(f"{what} {label!r} is not a real column (evaluates to {type(ar).__name__})")
    if ar.dtype.name not in kinds or not ar.dtype.isnative:
        raise _Decline(f"{what} {label!r} has dtype {ar.dtype}")
    return name, np.ascontiguousarray(ar)


def _binner_object_key(df, b, n_keys, run_pending=True):
    """a binner OBJECT passed as a key (round 6) -> (key column expression, what the result must look like), or _Decline.
    vaex.groupby.Grouper(expression, sort=, ascending=) over an integer column (vaex/groupby.py:226-330): the object has run its distinct-key
    pass already when it was made; the groups are its bin_values IN ITS ORDER (sorted either way, or the hash map's own; a missing-value
    group among them), typed the narrowest signed integer that holds them (no BinnerInteger simplification: allow_simplify is the wrapper's own, :599).
    vaex.groupby.BinnerInteger(expression[, min_value, max_value, dropmissing]) (:147-205): what df.groupby(<bool / int8 / uint8 column>) makes itself (:593-596).
    vaex.groupby.GrouperCategory(expression) (:384-442): the categories of a categorical column."""
    import vaex.array_types
    import vaex.groupby
    kind = type(b).__name__
    if type(b) is vaex.groupby.GrouperCategory:
        # the object's bins are the column's categories in the order IT settled on (natural, or sorted by label either way — pre_sort or not, its
        # bin_values are already in that order); alone it is `dense`: every category is a row of the result
        if b.df.dataset != df.dataset or getattr(b, "row_limit", None) is not None:
            raise _Decline(f"binner object as key ({kind} of another dataset / with a row limit)")
        return str(b.expression_original), {"kind": "category", "bin_values": b.bin_values}
    if type(b) is vaex.groupby.Grouper:
        if not hasattr(b, "hashmap_unique") and run_pending and getattr(getattr(b, "_promise", None), "isPending", False):
            # the object's distinct-key pass is scheduled, not run (vaex/groupby.py:298: delay=True; GroupBy.__init__ would run it now, :1021)
            b.df.execute()
        if getattr(b, "simpler", None) is not None or not hasattr(b, "bin_values") or not hasattr(b, "hashmap_unique"):
            raise _Decline(f"binner object as key ({kind} that has not run its distinct-key pass)")
        if b.df.dataset != df.dataset:
            raise _Decline(f"binner object as key ({kind} of another dataset)")
        bv, null_at = b.bin_values, None
        if hasattr(bv, "null_count"):   # (an arrow array: the sorted forms come back through pyarrow)
            if not (__import__("pyarrow").types.is_integer(bv.type)):
                raise _Decline(f"binner object as key ({kind} over {bv.type})")
            split = _split_nullable(bv) if bv.null_count else None
            bv = vaex.array_types.to_numpy(bv) if split is None else np.ma.array(split.data, mask=split.mask)
        if not isinstance(bv, np.ndarray) or bv.dtype.kind not in "iu" or getattr(b.hashmap_unique, "has_nan", False):
            raise _Decline(f"binner object as key ({kind} over {getattr(bv, 'dtype', type(bv).__name__)})")
        if np.ma.isMaskedArray(bv):
            at = np.flatnonzero(np.ma.getmaskarray(bv))
            if len(at) > 1:
                raise _Decline(f"binner object as key ({kind} with several missing-value groups)")
            null_at = int(at[0]) if len(at) else None
            bv = np.ma.getdata(bv) if null_at is None else bv
        if null_at is None and n_keys == 1 and not getattr(b.hashmap_unique, "has_null", False):
            return str(b.expression), {"kind": "grouper", "bin_values": np.asarray(bv)}   # (the plain road: _finish_arrays)
        return str(b.expression), {"kind": "bins", "values": np.ma.getdata(bv).astype(np.int64), "null_at": null_at, "bin_values": b.bin_values, "dense": True, "what": kind}
    if type(b) is vaex.groupby.BinnerInteger:
        if b.df.dataset != df.dataset:
            raise _Decline(f"binner object as key ({kind} of another dataset)")
        tiny = b.dtype.numpy.name in _TINY_KEYS
        if tiny and not getattr(b, "dropmissing", False) and n_keys == 1:
            return str(b.expression), {"kind": "integer", "invert": bool(b.invert)}   # (the plain road; a column with missing values leaves it in _plan)
        if b.dtype.numpy.kind not in "iu" or (b.invert and b.dtype.numpy.name == "bool"):
            raise _Decline(f"binner object as key ({kind} over {b.dtype})")
        values = np.arange(b.min_value, b.min_value + b.N, dtype=np.int64)
        values = values[::-1] if b.invert else values
        bv = b.bin_values
        if len(bv) != len(values) + (0 if b.dropmissing else 1):
            raise _Decline(f"binner object as key ({kind} whose bins are not min_value .. max_value)")
        # (BinnerOrdinal counts a row outside min_value .. max_value into the MISSING-VALUE bin — src/binner_ordinal.cpp:70-72, "negative values are
        #  interpreted as null, as well as out of bound" — which a BinnerInteger keeps unless dropmissing: "outside" tells _coded_key to do the same)
        return str(b.expression), {"kind": "bins", "values": values, "null_at": None if b.dropmissing else len(values), "bin_values": bv, "dense": bool(b.dense), "what": kind,
                                   "outside": (int(b.min_value), int(b.min_value) + int(b.N) - 1)}
    if type(b) is vaex.groupby.BinnerTime:
        # vaex.BinnerTime(expression, resolution, every) (vaex/groupby.py:63-144): bin = (t in the resolution's units - tmin in those units) // every over
        # N bins from the column's own minmax; not dense — only bins with a row are rows of the result, labelled with the bin's first instant
        if b.df.dataset != df.dataset:
            raise _Decline(f"binner object as key ({kind} of another dataset)")
        unit, every, t0, count = b.resolution_type, int(b.every), b.tmin, int(b.N)
        if len(b.bin_values) != count:
            raise _Decline(f"binner object as key ({kind} whose bins are not tmin .. tmax)")

        def codes_of(t):
            steps = (t.astype(unit) - t0.astype(unit)).astype(np.int64)
            steps[np.isnat(t)] = -1
            return np.floor_divide(steps, every)
        return str(b.expression), {"kind": "bins", "values": np.arange(count, dtype=np.int64), "null_at": None, "bin_values": b.bin_values, "dense": False, "what": kind, "codes_of": codes_of}
    raise _Decline(f"binner object as key ({kind})")


def _materialised(df, name, what):
    """the whole real column `name` (held as arrow / behind a dataset proxy) as ONE numpy array over the active range — for keys whose codes are
    made on the host (float keys); frames up to materialise_max_rows rows"""
    if len(df) > materialise_max_rows or df.length_original() > materialise_max_rows:
        raise _Decline(f"{what} {name!r} is not one array")
    try:
        import vaex.array_types
        ar = df.evaluate(name, filtered=False, parallel=False)
        ar = ar if isinstance(ar, np.ndarray) else np.asarray(vaex.array_types.to_numpy(ar))
    except Exception as e:   # noqa: BLE001
        raise _Decline(f"{what} {name!r} is not one array ({type(e).__name__})")
    if np.ma.isMaskedArray(ar) or ar.ndim != 1:
        raise _Decline(f"{what} {name!r} is not one array")
    return np.ascontiguousarray(ar if ar.dtype.isnative else ar.astype(ar.dtype.newbyteorder("=")))


def _datetime_column(df, name, what):
    """the numpy datetime64 array behind the real column `name` (active range applied), or _Decline"""
    ar = df.columns.get(name)
    if not isinstance(ar, np.ndarray) or np.ma.isMaskedArray(ar) or ar.dtype.kind != "M" or ar.ndim != 1 or not ar.dtype.isnative or len(ar) > materialise_max_rows:
        raise _Decline(f"{what} {name!r} is not a plain numpy datetime column")
    i1, i2 = df._index_start, df._index_end
    return ar[i1:len(ar) if i2 is None else i2]


def _category_key(df, name, sort, ascending, bin_values=None):
    """a categorical key column (df.categorize / ordinal_encode: integer codes min_value .. min_value + N - 1 with N labels; vaex's GrouperCategory,
    vaex/groupby.py:384-442) -> how the device's groups — the CODES that occur, ascending — become the rows vaex hands back: the labels in
    natural order or sorted by label (`order[p]` = label index of bin p), codes outside the categories dropped (BinnerOrdinal's overflow bin,
    extract_center), and — a single key is `dense` — a row for every category, with or without rows"""
    import pyarrow as pa
    import vaex.array_types
    try:
        labels = df.category_labels(name, aslist=False)
        count, offset = int(df.category_count(name)), int(df.category_offset(name))
    except Exception as e:   # noqa: BLE001
        raise _Decline(f"group key {name!r} is categorical ({type(e).__name__})")
    labels = pa.array(labels) if isinstance(labels, list) else labels
    if len(labels) != count:
        raise _Decline(f"group key {name!r} is categorical (labels and count disagree)")
    natural = np.arange(count, dtype=np.int64)
    if bin_values is not None:   # a GrouperCategory OBJECT: which of the three orders did it take?
        mine = vaex.array_types.to_arrow(bin_values)
        mine = pa.concat_arrays(mine.chunks) if isinstance(mine, pa.ChunkedArray) else mine
        ref = vaex.array_types.to_arrow(labels)
        ref = pa.concat_arrays(ref.chunks) if isinstance(ref, pa.ChunkedArray) else ref
        for order in (natural, None, False):
            if order is not natural:
                order = vaex.array_types.to_numpy(pa.compute.sort_indices(ref, sort_keys=[("x", "ascending" if order is None else "descending")])).astype(np.int64)
            if mine.equals(ref.take(pa.array(order))):
                break
        else:
            raise _Decline("binner object as key (GrouperCategory whose bins are not the column's categories)")
        bins = bin_values
    elif sort:
        order = vaex.array_types.to_numpy(pa.compute.sort_indices(labels, sort_keys=[("x", "ascending" if ascending else "descending")])).astype(np.int64)
        bins = pa.compute.take(labels, pa.array(order))
    else:
        order, bins = natural, labels
    rank = np.empty(count, dtype=np.int64)
    rank[order] = natural
    return {"kind": "category", "offset": offset, "count": count, "bins": bins, "rank": rank}


#: from this many rows on, the codes of a key with missing values / of a float key, and the NaN-filled form of a value column with missing entries, are made
#: ON THE DEVICE (vxh_code_column behind a multi-threaded upload of data and mask) instead of by numpy passes over the host column — on the GPU box's
#: host one numpy thread codes 1e8 int64 keys in ~0.35 s, more than vaex's own two passes take on 256 threads (profiles/r06_keykinds_timing.txt)
device_coding_min_rows = 2_000_000
_FLOAT_NAN_CODE, _FLOAT_NULL_CODE = 0x7ff8000000000000, 0x7ff8000000000001   # NaN bit patterns no value has once every NaN is the first of them


class _Lazy:
    """a column of the plan that is MADE from host arrays — the int64 codes of a key (`mode` "key"; `meta` is the key's entry of plan.key_meta and gets its
    null_code when the codes exist), or float64 values with NaN where an entry is missing ("value") — and not made yet.  The real _frame_for makes it on
    the device (device()); host() is the numpy road (small columns never become a _Lazy: device_coding_min_rows)."""

    def __init__(self, mode, name, data, mask, meta=None, exact_sums=False):
        self.mode, self.name, self.data, self.mask, self.meta, self.exact_sums = mode, name, data, mask, meta, exact_sums
        self.dtype = np.dtype(np.int64 if mode == "key" else np.float64)

    def __len__(self):
        return len(self.data)

    def host(self):
        if self.mode == "value":
            return _nan_filled_host(self.name, self.data, self.mask, self.exact_sums)
        return _codes_host(self.name, self.data, self.mask, self.meta)

    def device(self, sa, torch, data, mask):
        """`data` / `mask`: the host arrays' device copies (torch tensors; bool as uint8) -> (torch tensor of the made column, its owner)"""
        kind = "uint8" if self.data.dtype.kind == "b" else self.data.dtype.name
        dt = binned._DT_CODE[kind]
        n = len(self.data)
        if self.mode == "value":
            if self.exact_sums and n:
                lo, hi = sa.minmax_int(data, None, dt, False)
                if float(max(abs(int(lo)), abs(int(hi)))) * n >= 2.0 ** 53:
                    raise _Decline(f"aggregated expression {self.name!r}: integer column with missing values too large for exact float64 sums")
            out = sa.code_column(data, mask, dt, 1)
        elif self.data.dtype.kind == "f":
            out = sa.code_column(data, mask, dt, 0, _FLOAT_NULL_CODE, _FLOAT_NAN_CODE)
        else:
            # (the missing rows' code: one past the largest element — what the column holds UNDER its mask counts too: a code no value has, which is all it must be)
            lo, hi = sa.minmax_int(data, None, dt, False) if n else (0, 0)
            if int(hi) >= np.iinfo(np.int64).max:
                raise _Decline(f"group key {self.name!r}: no code left for the missing values")
            self.meta["null_code"] = int(hi) + 1
            out = sa.code_column(data, mask, dt, 0, int(hi) + 1, 0)
        return torch.as_tensor(out, device=data.device), out


def _codes_host(name, data, mask, meta):
    """int64 codes of a by-name key with numpy (see _coded_key); sets meta["null_code"]"""
    if data.dtype.kind == "f":
        v = data.astype(np.float64)   # (a copy of the call's own: its bits become the codes in place)
        codes = v.view(np.int64)
        nan = np.isnan(v)
        if nan.any():
            codes[nan] = _FLOAT_NAN_CODE
        if mask is not None:
            codes[mask] = _FLOAT_NULL_CODE
        return codes
    codes = data.astype(np.int64)
    top = int(np.max(codes, where=~mask, initial=np.iinfo(np.int64).min)) if mask is not None and not mask.all() else (int(codes.max()) if mask is None and len(codes) else 0)
    if top >= np.iinfo(np.int64).max:
        raise _Decline(f"group key {name!r}: no code left for the missing values")
    meta["null_code"] = top + 1
    if mask is not None:
        codes[mask] = top + 1
    return codes


def _nan_filled_host(name, data, mask, exact_sums):
    if exact_sums:
        live = data[~mask]
        if len(live) and float(np.abs(live.astype(np.float64)).max()) * len(data) >= 2.0 ** 53:
            raise _Decline(f"aggregated expression {name!r}: integer column with missing values too large for exact float64 sums")
    out = data.astype(np.float64)
    out[mask] = np.nan
    return out


def _coded_key(name, ar, obj, key_object, sort, ascending):
    """(int64 codes the device groups, the key's meta for _finish_general) of a key column with missing values (`ar` a _Nullable) and / or under a
    binner object with enumerated bins (`obj`, kind "bins").  The missing rows' code is one past the largest value of the column and of the bins."""
    if isinstance(ar, _Nullable):
        data, mask = ar.data, ar.mask
    else:
        data, mask = np.asarray(ar), None
    source = data.dtype.name
    lazy = len(data) >= device_coding_min_rows and obj is None
    if data.dtype.kind == "f":
        # a float key (vaex's Grouper over ordered_set<double>: one group per value, one for NaN, one for the missing values, vaex/groupby.py:226-330):
        # the device groups the BIT PATTERNS of the float64 values (-0.0 and 0.0 are two keys, as in the reference's hash map: src/hash_primitives.hpp
        # compares what it hashes); every NaN becomes one NaN pattern, a missing value another — patterns no value has
        meta = {"kind": "coded", "float": True, "null_code": _FLOAT_NULL_CODE, "nan_code": _FLOAT_NAN_CODE, "source": source, "sort": bool(sort), "ascending": bool(ascending)}
        return (_Lazy("key", name, data, mask, meta) if lazy else _codes_host(name, data, mask, meta)), meta
    if data.dtype.kind not in "biu":
        raise _Decline(f"group key {name!r} has dtype {data.dtype}")
    if obj is None:   # a key by name (or a tiny BinnerInteger over a column with missing values): ordered and typed by _finish_general
        meta = {"kind": "coded", "null_code": None, "source": source}
        if key_object is not None and key_object["kind"] == "integer":   # BinnerInteger over bool / int8 / uint8: ascending or inverted, the missing values last
            meta.update(sort=True, ascending=not key_object["invert"], tiny=True)
        else:
            meta.update(sort=bool(sort), ascending=bool(ascending), tiny=source in _TINY_KEYS)
        return (_Lazy("key", name, data, mask, meta) if lazy else _codes_host(name, data, mask, meta)), meta
    codes = data.astype(np.int64)
    top = int(np.max(codes, where=~mask, initial=np.iinfo(np.int64).min)) if mask is not None and not mask.all() else (int(codes.max()) if mask is None and len(codes) else 0)
    if obj is not None and len(obj["values"]):
        top = max(top, int(obj["values"].max()))
    if top >= np.iinfo(np.int64).max:
        raise _Decline(f"group key {name!r}: no code left for the missing values")
    null_code = top + 1
    if mask is not None:
        codes[mask] = null_code
    if obj is not None and obj.get("outside") is not None:
        codes[(codes < obj["outside"][0]) | (codes > obj["outside"][1])] = null_code
    meta = {"kind": "coded", "null_code": null_code, "source": source}
    if obj is not None:   # enumerated bins, in the object's order
        in_order = obj["values"].copy()
        if obj["null_at"] is not None:
            in_order = in_order if obj["null_at"] < len(in_order) else np.append(in_order, 0)
            in_order[obj["null_at"]] = null_code
        meta.update(in_order=in_order, bins=obj["bin_values"], dense=bool(obj["dense"]), what=obj["what"])
    elif key_object is not None and key_object["kind"] == "integer":   # BinnerInteger over bool / int8 / uint8: ascending or inverted, the missing values last
        meta.update(sort=True, ascending=not key_object["invert"], tiny=True)
    else:
        meta.update(sort=bool(sort), ascending=bool(ascending), tiny=source in _TINY_KEYS)
    return codes, meta


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
    leaf = _translate(df, aggregate, columns, predicates)
    if isinstance(leaf, dict):
        raise _Decline("nunique inside an aggregator expression")
    spec[name] = leaf
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
    if isinstance(aggregate, vaex.agg.AggregatorDescriptorBasic) and aggregate.name == "AggNUnique" and selection is None and len(aggregate.expressions) == 1 and isinstance(columns, _Columns):
        # round 6: nunique(x) per group = the number of distinct (keys, x) combinations per group — a second device groupby over the keys and x
        # (_nunique_pass); integer / bool x, missing values a value of their own unless dropmissing (src/agg_nunique.cpp)
        name, ar = _real_column(df, aggregate.expressions[0], _KEY_KINDS, "nunique expression", nullable=True)
        nu = {"nunique": True, "dropmissing": bool(aggregate.dropmissing), "meta": None}
        if isinstance(ar, _Nullable):
            if aggregate.dropmissing:
                raise _Decline("nunique(dropmissing=True) of a column with missing values")   # (the reference's own answer there is not a count: INTEGRATION.md "Differences")
            alias = "__codes_of_" + name
            if alias not in columns:
                columns[alias], meta = _coded_key(name, ar, None, None, False, True)
                columns.host_made.add(alias)
                columns.null_codes[alias] = meta   # (its "null_code" is known when the codes are made: _Lazy)
            nu.update(column=alias, meta=columns.null_codes.get(alias))
        else:
            columns.setdefault(name, ar)
            nu.update(column=name)
        return nu
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
    name, ar = _real_column(df, expressions[0], _VALUE_KINDS, "aggregated expression", nullable=True)
    if isinstance(ar, _Nullable):
        ar = _nullable_values(name, ar, kind, columns)
    if selection is not None and kind in ("min", "max"):
        raise _Decline("min / max with a selection")   # (a group without a selected row: vaex hands back the dtype's extreme, masked or not by dtype)
    columns[name] = ar   # (var / std of an integer column: the primitives run on astype(float64), as vaex/agg.py:427 does)
    return getattr(binned.agg, kind)(name, selection=selection)


class _Columns(dict):
    """the plan's columns; `int_nullable`: integer value columns with missing values that ride the pass as float64 with NaN (their sums are integers again)"""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.int_nullable = set()
        self.null_codes = {}     # coded column -> the code of its missing values
        self.host_made = set()   # columns whose rows were MADE on the host for this call (codes of a key, NaN for missing values): not what the executor's chunks hold


def _nullable_values(name, ar, kind, columns):
    """a value column with missing values (round 6) -> float64 with NaN where a value is missing: count / sum / the moments skip NaN exactly as they skip
    a missing value (src/agg_count.cpp:56, agg_sum.cpp:113).  An integer column only while every partial sum stays exact in float64 (rows x largest
    magnitude < 2^53); its sums are handed back as int64.  min / max keep their dtype and mask in vaex: not taken."""
    if kind in ("min", "max"):
        raise _Decline(f"min / max of {name!r}, a column with missing values")
    if name in columns and not isinstance(columns[name], (np.ndarray, _Lazy)):
        raise _Decline(f"aggregated expression {name!r} is also a key with missing values")
    if name in columns:
        return columns[name]   # (converted for another aggregation of this call)
    data, mask = ar.data, ar.mask
    exact_sums = data.dtype.kind in "iu"
    if exact_sums:
        if not hasattr(columns, "int_nullable"):
            raise _Decline(f"aggregated expression {name!r} is an integer column with missing values")
        columns.int_nullable.add(name)
    elif data.dtype.kind != "f":
        raise _Decline(f"aggregated expression {name!r} has dtype {data.dtype}")
    if hasattr(columns, "host_made"):
        columns.host_made.add(name)
    if len(data) >= device_coding_min_rows:
        return _Lazy("value", name, data, mask, exact_sums=exact_sums)
    return _nan_filled_host(name, data, mask, exact_sums)


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


def _key_column_like_vaex(values, source_kind=None, has_null=False):
    """a key column of the result typed the way vaex's groupers hand it back (vaex/groupby.py:147-205, :263-277) — decided per key
    from its distinct values: BinnerInteger (range <= 4/3 of the distinct keys) -> int64 with an (empty) mask; else Grouper -> the
    narrowest signed integer type that holds the range.  bool / int8 / uint8 keys are BinnerInteger from the start (vaex/groupby.py:
    593-595): bin_values [False, True, null] resp. arange + null as a masked int64 array (:166-187).  `has_null`: the key has a group of
    missing values — the reference's hash map hands that key back among its `bin_values`, so `bins = len(self.bin_values)` (:266) counts it."""
    k = np.asarray(values)
    if source_kind == "bool":
        return np.ma.array(k.astype(bool), mask=np.zeros(len(k), dtype=bool), shrink=False)
    if source_kind in ("int8", "uint8"):
        return np.ma.array(k.astype(np.int64), mask=np.zeros(len(k), dtype=bool), shrink=False)
    if len(k) == 0:   # (no group: vaex hands back an empty column of the key's own type)
        return k.astype(source_kind) if source_kind else k
    vmin, vmax = int(k.min()), int(k.max())
    distinct = (len(k) if np.all(k[1:] > k[:-1]) or np.all(k[1:] < k[:-1]) else len(np.unique(k))) + (1 if has_null else 0)
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
    __slots__ = ("by", "agg", "sort", "srt", "asc", "columns", "key_names", "actions", "spec", "selection", "rows", "predicates", "streamed", "key_object", "finishers", "key_meta", "key_label", "nunique")


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
    key_object, objects = None, {}
    for i, b in enumerate(by_list):
        if isinstance(b, vaex.groupby.BinnerBase):
            by_list[i], obj = _binner_object_key(df, b, len(by_list), run_pending=not for_task)
            if obj["kind"] in ("category", "bins"):
                objects[i] = obj
            else:
                key_object = obj
    asc = list(ascending) if isinstance(ascending, (list, tuple)) else [ascending] * len(by_list)
    srt = list(sort) if isinstance(sort, (list, tuple)) else [sort] * len(by_list)
    if len(asc) != len(by_list) or len(srt) != len(by_list):
        raise _Decline("sort / ascending lists of another length than the keys")
    if key_object is not None:   # (the object's own order, not the call's: vaex/groupby.py:632-636 passes sort / ascending to the groupers IT makes)
        srt, asc = ([True], [not key_object["invert"]]) if key_object["kind"] == "integer" else ([False], [True])
    columns, key_names, key_meta, key_originals, key_label = _Columns(), [], {}, [], {}
    for i, b in enumerate(by_list):
        obj = objects.get(i)
        if obj is not None and "codes_of" in obj:   # (BinnerTime: the bin index of every row, computed once on the host from the datetime column)
            name = str(vaex.utils._ensure_string_from_expression(b))
            ar = obj["codes_of"](_datetime_column(df, name, "group key"))
        else:
            name, ar = _real_column(df, vaex.utils._ensure_string_from_expression(b), _KEY_KINDS + _FLOAT_KEYS, "group key", nullable=True)
        if name in columns or name in key_originals:
            raise _Decline("the same key twice")
        if ar.dtype.name in _FLOAT_KEYS and (obj is not None or key_object is not None):
            raise _Decline(f"group key {name!r} has dtype {ar.dtype} under a binner object")
        if ar.dtype.name in _FLOAT_KEYS and isinstance(ar, _Streamed):
            ar = _materialised(df, name, "group key")   # (the codes are made from the whole column: arrow without nulls is a view, a proxied column one evaluate)
        if ((obj is not None and obj["kind"] == "category") or (obj is None and df.is_category(name))) and not (key_object is not None and key_object["kind"] == "grouper"):   # (a Grouper OBJECT groups the codes like any integer column)
            # round 6: vaex's GrouperCategory (vaex/groupby.py:384-442; chosen before the dtype is looked at, :587-589) — the device groups the
            # codes, _finish_general turns them into the labels, in the grouper's order, one row per category when the key stands alone
            if isinstance(ar, _Nullable):
                raise _Decline(f"group key {name!r} is categorical with missing codes")
            key_meta[name] = _category_key(df, name, srt[i], asc[i], bin_values=None if obj is None else obj["bin_values"])
        if key_object is not None and key_object["kind"] == "integer" and ar.dtype.name not in _TINY_KEYS:
            raise _Decline("binner object as key (BinnerInteger)")
        if ar.dtype.name in _TINY_KEYS and len(by_list) > 1 and name not in key_meta and obj is None:
            raise _Decline(f"{ar.dtype.name} key next to other keys")   # (BinnerInteger's N is the dtype's range, not the distinct keys: vaex's combine decision differs)
        if isinstance(ar, _Nullable) or (obj is not None and obj["kind"] == "bins") or ar.dtype.name in _FLOAT_KEYS:
            # round 6: a key with missing values, and / or a binner object whose bins are an enumeration (a Grouper with a missing-value group or
            # next to other keys, a BinnerInteger over a range): the device groups int64 CODES — the values, the missing rows under a code of their own
            if key_object is not None and key_object["kind"] == "grouper":
                raise _Decline("binner object as key (Grouper over a column with missing values that it does not list)")
            if isinstance(ar, _Streamed):
                ar = _materialised(df, name, "group key")   # (the codes are made from the whole column: arrow without nulls is a view, a proxied column one evaluate)
            ar, meta = _coded_key(name, ar, obj, key_object, srt[i], asc[i])
            key_originals.append(name)
            key_label["__codes_of_" + name] = name   # (the codes live beside the column itself: a filter or an aggregation may read that too)
            name = "__codes_of_" + name
            key_meta[name] = meta
            columns.host_made.add(name)
            if key_object is not None:   # (a tiny BinnerInteger over a column with missing values: the general road orders and types it)
                key_object = None
        else:
            key_originals.append(name)
        columns[name] = ar
        key_names.append(name)
    if not key_meta and len(set(zip(srt, asc))) > 1:
        raise _Decline("keys sorted in different directions")   # (_finish_general orders key by key; the plain road reverses the whole result)
    actions = _normalise_actions(df, key_originals, agg)
    if not actions:
        raise _Decline("no aggregation")
    spec, predicates, finishers, nunique = {}, {}, {}, {}
    for out_name, aggregate in actions:
        if out_name in spec or out_name in finishers or out_name in key_originals or out_name in nunique:
            raise _Decline("duplicate output column")
        if isinstance(aggregate, _expression_types()):
            # arithmetic over aggregators (vaex.agg.sum('x') / vaex.agg.count(), -vaex.agg.mean('y'), ...: vaex/agg.py:77-189): the leaves are
            # aggregations of the same pass under hidden names, the operators run over their per-group columns when the groups exist
            finishers[out_name] = _translate_tree(df, aggregate, columns, predicates, spec)
        else:
            one = _translate(df, aggregate, columns, predicates)
            if isinstance(one, dict):
                if one["column"] in key_names:
                    raise _Decline("nunique of a key")
                nunique[out_name] = one
            else:
                spec[out_name] = one
    # a filtered frame (df[df.x > 0].groupby(...)): vaex compacts every chunk of every column with numpy before its two passes see a row
    # (vaex/execution.py:515-523); here the filter is a device predicate in every aggregator's keep-mask (vaex_amd/vaex_filter.py) and
    # groups without a row inside it are dropped — when it is in the predicate subset over real numeric columns; else vaex's own code
    selection = None
    if df.filtered and not for_task:
        from . import vaex_filter
        pred = vaex_filter.filter_plan(df)
        if pred is None and vaex_filter.filter_expression(df) is not None:
            # (round 6: a filter over byte-swapped numeric columns — the executor's chunk predicates want native columns; this call converts such a
            #  column once on the host, _real_column — compiled against stand-ins of the native dtype)
            from . import predicate, vaex_selection
            known = dict(vaex_selection._known_columns(df))
            for cname, car in df.columns.items():
                if cname not in known and isinstance(car, np.ndarray) and not np.ma.isMaskedArray(car) and car.ndim == 1 and not car.dtype.isnative and car.dtype.kind in "iuf":
                    known[cname] = np.empty(0, dtype=car.dtype.newbyteorder("="))
            try:
                pred = predicate.compile_selection(vaex_filter.filter_expression(df), known, virtual=vaex_selection._virtual_columns(df))
            except predicate.Unsupported:
                pred = None
        if pred is None:
            raise _Decline("filtered DataFrame (filter outside the device predicate subset)")
        selection = vaex_filter.filter_expression(df)
        predicates[selection] = pred
        for c in pred.columns:
            if c not in columns:
                columns[c] = _real_column(df, c, tuple(k for k in vaex_filter._NUMERIC if k != "bool"), "filter column", nullable=True)[1]
            if not isinstance(columns[c], (np.ndarray, _Streamed)):
                raise _Decline(f"filter column {c!r} has missing values")
    if columns.host_made:
        # (a column made on the host is not what the executor would evaluate chunk by chunk: such a call is eager over whole arrays — its other
        #  columns held in arrow / behind a proxy are evaluated once — and not a task of a delayed pass)
        if for_task:
            raise _Decline("delayed groupby over a key or value column made on the host (missing values, float keys, binner objects)")
        for cname in list(columns):
            if isinstance(columns[cname], _Streamed):
                columns[cname] = _materialised(df, cname, "column")
    plan = _Plan()
    plan.by, plan.agg, plan.sort, plan.srt, plan.asc = by, agg, sort, srt, asc
    plan.columns, plan.key_names, plan.actions, plan.spec, plan.selection = columns, key_names, actions, spec, selection
    plan.rows = len(next(iter(columns.values())))
    plan.predicates = predicates
    plan.streamed = any(isinstance(c, _Streamed) for c in columns.values())
    plan.key_object = key_object
    plan.finishers = finishers
    plan.key_meta = key_meta
    plan.key_label = key_label
    plan.nunique = nunique
    if key_meta and len(key_names) == 1:   # a dense key: a bin without a row is a row of the result — count 0, sum 0, mean / var / std NaN; min / max: vaex's own
        m0 = key_meta[key_names[0]]
        for d in spec.values():
            if d.name in ("min", "max") and (m0["kind"] == "category" or m0.get("dense", False)):
                raise _Decline("min / max over a dense enumerated key")

    return plan


def _run(plan, frame):
    """the groups of `frame` (a binned.Frame over the plan's columns): {column: array}, or _Decline"""
    key_names = plan.key_names
    try:
        frame.last_groupby_info = None
        frame._predicates.update(plan.predicates)   # (compiled against the DataFrame: virtual columns are inlined there)
        spec = dict(plan.spec) if plan.spec or not plan.nunique else {"__rows__": binned.agg.count()}
        res = frame.groupby(key_names if len(key_names) > 1 else key_names[0], spec, selection=plan.selection)
        if plan.nunique:
            info = frame.last_groupby_info
            res = dict(res)
            for out_name, nu in plan.nunique.items():
                res[out_name] = _nunique_pass(plan, frame, res, nu)
            frame.last_groupby_info = dict(info or {}, nunique_passes=len(plan.nunique))
        return res
    except (NotImplementedError, ValueError) as e:
        raise _Decline(str(e))
    except (RuntimeError, MemoryError) as e:
        # a failure the device path reports (HBM exhausted by the device copies or the partition queues, a HIP error): vaex's own two
        # passes answer — chunk by chunk, on the HIP classes where those still work — instead of the call dying here
        drop_device_copies()
        raise _Decline(f"device groupby failed: {type(e).__name__}: {str(e)[:200]}")


def _nunique_pass(plan, frame, res, nu):
    """nunique(x) per group of `res` (the main pass's groups, ascending by the keys): ONE more device groupby over (keys..., x) — its groups are the
    distinct combinations — and a run-length count per key combination on the host (as many entries as there are distinct combinations)"""
    key_names = plan.key_names
    pairs = frame.groupby(list(key_names) + [nu["column"]], {"__pairs__": binned.agg.count()}, selection=plan.selection)
    x = np.asarray(pairs[nu["column"]]).astype(np.int64)
    keys = [np.asarray(pairs[k]).astype(np.int64) for k in key_names]
    null_code = (nu.get("meta") or {}).get("null_code")
    if nu["dropmissing"] and null_code is not None:
        live = x != null_code
        keys = [k[live] for k in keys]
    stacked = np.stack(keys, axis=1) if len(keys[0]) else np.zeros((0, len(keys)), dtype=np.int64)
    main = np.stack([np.asarray(res[k]).astype(np.int64) for k in key_names], axis=1)
    out = np.zeros(len(main), dtype=np.int64)
    if len(stacked):
        start = np.ones(len(stacked), dtype=bool)
        start[1:] = np.any(stacked[1:] != stacked[:-1], axis=1)
        at = np.flatnonzero(start)
        counts = np.diff(np.append(at, len(stacked)))
        both, inverse = np.unique(np.concatenate([main, stacked[at]]), axis=0, return_inverse=True)
        inverse = inverse.reshape(-1)
        where = np.full(len(both), -1, dtype=np.int64)
        where[inverse[:len(main)]] = np.arange(len(main))
        pos = where[inverse[len(main):]]
        if (pos < 0).any():
            raise _Decline("nunique: the pair pass found a group the main pass did not")
        out[pos] = counts
    return out


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
    if getattr(columns, "int_nullable", None):
        # (an integer column with missing values rode the pass as float64 with NaN, _nullable_values: its sums are integers again)
        res = dict(res)
        for name, d in plan.spec.items():
            if d.name == "sum" and d.column in columns.int_nullable:
                res[name] = np.rint(np.asarray(res[name])).astype(np.int64)
    if plan.key_meta:
        return _finish_general(df, plan, frame, res)
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


def _finish_general(df, plan, frame, res):
    """_finish_arrays where a key is not handed back as the integers the device grouped (round 6: categorical keys, keys with missing values,
    binner objects with enumerated bins).  The device's groups — one row per combination of key CODES that occurs — are filtered (codes outside
    a key's bins: BinnerOrdinal's overflow bin, vaex/groupby.py extract_center), ordered key by key (first key slowest, each in ITS grouper's bin
    order: vaex's grid order, vaex/groupby.py:941-948; missing values last, as arrow sorts them), a single dense key gets a row for every bin
    (count 0, sum 0, mean / var / std NaN: what vaex's finishers make of an empty cell), and the keys are decoded (labels, masked arrays)."""
    import pyarrow as pa
    key_names, columns, actions, metas = plan.key_names, plan.columns, plan.actions, plan.key_meta
    codes = {name: np.asarray(res[name]).astype(np.int64) for name in key_names}
    n = len(codes[key_names[0]])
    for name in key_names:
        # a key by name with a group of missing values: the reference's Grouper counts that group among its bins when it decides whether the values are a
        # dense integer range (vaex/groupby.py:263-272: `bins = len(self.bin_values)`, `dense = bins == int_range`).  Two corners follow that are left to
        # the reference itself (seen once each in 10000 random calls): every row missing — `bin_values.min()` is the masked constant and BinnerInteger raises
        # "Cannot store the range" — and a range with exactly ONE integer no row has: null + values == range, the lone key is "dense" and the absent integer
        # is handed back as a group without a row (count 0, mean NaN, min inf)
        m = metas.get(name)
        if m is not None and m["kind"] == "coded" and "in_order" not in m and not m.get("float") and not m.get("tiny") and n:
            real = codes[name] != m["null_code"]
            if not real.all():
                if not real.any():
                    raise _Decline(f"group key {name!r}: every row is missing (the reference raises)")
                u = np.unique(codes[name][real])
                if len(key_names) == 1 and int(u[-1]) - int(u[0]) + 1 == len(u) + 1:
                    raise _Decline(f"group key {name!r}: missing values next to a range with one absent integer (the reference hands that integer back as a group without a row)")
    keep = np.ones(n, dtype=bool)
    at = {}
    for name in key_names:
        m, c = metas.get(name), codes[name]
        if m is None:
            continue
        if m["kind"] == "category":
            keep &= (c >= m["offset"]) & (c < m["offset"] + m["count"])
        elif "in_order" in m:
            o = np.argsort(m["in_order"], kind="stable")
            sc = m["in_order"][o]
            pos = np.minimum(np.searchsorted(sc, c), max(len(sc) - 1, 0))
            found = (sc[pos] == c) if len(sc) else np.zeros(n, dtype=bool)
            keep &= found
            at[name] = o[pos] if len(sc) else np.zeros(n, dtype=np.int64)
    ranks, cells, sizes = [], 1, []
    for i, name in enumerate(key_names):
        m = metas.get(name)
        c = codes[name][keep]
        if m is not None and m["kind"] == "category":
            ranks.append(m["rank"][c - m["offset"]])
            sizes.append(m["count"])
        elif m is not None and "in_order" in m:
            ranks.append(at[name][keep])
            sizes.append(len(m["in_order"]))
        else:
            down = (m["sort"] and not m["ascending"]) if m is not None else (plan.srt[i] and not plan.asc[i])
            if m is not None and m.get("float"):   # by value; NaN behind the numbers, the missing values last (arrow's order either way)
                v = c.view(np.float64).copy()
                cls = np.where(c == m["null_code"], 2, np.where(c == m["nan_code"], 1, 0))
                v[cls > 0] = 0.0
                ranks.append(cls)
                ranks.append(-v if down else v)
                ranks.append(c)   # (-0.0 and 0.0 compare equal: the bit pattern settles their order)
            else:
                r = -c if down else c.copy()
                if m is not None:
                    cls = (c == m["null_code"]).astype(np.int64)
                    r[cls > 0] = 0
                    ranks.append(cls)
                ranks.append(r)
            sizes.append(len(np.unique(c)))
        cells *= max(1, sizes[-1])
    order = np.lexsort(ranks[::-1]) if n else np.arange(0)
    rank_of = {}   # (key index -> its position among `ranks`: enumerated keys contribute one array)
    j = 0
    for i, name in enumerate(key_names):
        m = metas.get(name)
        enumerated = m is not None and (m["kind"] == "category" or "in_order" in m)
        rank_of[i] = j
        j += 1 if (enumerated or m is None) else (3 if m.get("float") else 2)
    pick = np.flatnonzero(keep)[order]
    m0 = metas.get(key_names[0])
    dense = len(key_names) == 1 and m0 is not None and (m0["kind"] == "category" or m0.get("dense", False))
    combined = len(key_names) >= 2 and plan.rows / cells < 10
    values = {}
    for name, d in list(plan.spec.items()) + [(name, binned.agg.count()) for name in plan.nunique]:
        col = np.asarray(res[name])[pick]
        if dense:
            if d.name == "count":
                full = np.zeros(sizes[0], dtype=col.dtype if len(col) else np.int64)
            elif d.name == "sum":
                src = columns[d.column].dtype.kind if d.column in columns else "f"   # (no group at all: the sum's type follows the column's, upcast<>: src/agg_sum.cpp:6-62)
                empty = np.int64 if (src in "ib" or d.column in getattr(columns, "int_nullable", ())) else (np.uint64 if src == "u" else np.float64)
                full = np.zeros(sizes[0], dtype=col.dtype if len(col) else empty)
            else:
                full = np.full(sizes[0], np.nan, dtype=np.float64)
            full[ranks[rank_of[0]][order]] = col
            col = full
        values[name] = col
    out = {}
    for i, name in enumerate(key_names):
        m = metas.get(name)
        if m is not None and (m["kind"] == "category" or "in_order" in m):
            bins = m["bins"]
            if dense:
                k = bins
            else:
                idx = ranks[rank_of[i]][order]
                k = bins.take(pa.array(idx)) if hasattr(bins, "null_count") else (bins[idx] if isinstance(bins, np.ndarray) else np.asarray(bins)[idx])
        elif m is not None and m.get("float"):
            c = codes[name][pick]
            k = c.view(np.float64).astype(m["source"])
            if (c == m["null_code"]).any():
                k = np.ma.array(k, mask=c == m["null_code"], shrink=False)
        elif m is not None:
            c = codes[name][pick]
            real = c != m["null_code"]
            typed = _key_column_like_vaex(c[real], m["source"], has_null=not real.all() and not m.get("tiny"))
            if real.all():
                k = np.ma.getdata(typed) if combined else typed
            else:
                data = np.zeros(len(c), dtype=np.ma.getdata(typed).dtype)
                data[real] = np.ma.getdata(typed)
                k = np.ma.array(data, mask=~real, shrink=False)
        else:
            k = _key_column_like_vaex(codes[name][pick], columns[name].dtype.name)
            k = np.ma.getdata(k) if combined else k
        out[df[plan.key_label.get(name, name)]._label] = k
    for out_name, _ in actions:
        out[out_name] = _finish_tree(plan.finishers[out_name], values) if out_name in plan.finishers else values[out_name]
    last.clear()
    fused = frame.last_groupby_info and not frame.last_groupby_info.get("dense")
    ran = sorted({frame.sa.last_kernel(t) for t in getattr(frame, "last_slots", [0])} - {""}) if hasattr(frame.sa, "last_kernel") else []
    last.update(path="device", kernel="gb_scatter+gb_reduce" if fused else "+".join(ran), info=frame.last_groupby_info, groups=len(next(iter(out.values()))))
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


def _made_columns(columns):
    """the plan's columns with every _Lazy MADE: on the device — every column of the call uploaded with several copy threads, the lazy ones coded there by
    vxh_code_column — when the call's columns are plain numeric arrays and fit; else with numpy on the host.  -> (columns, owners of device arrays)"""
    lazies = [name for name, c in columns.items() if isinstance(c, _Lazy)]
    if not lazies:
        return columns, []
    try:
        import torch
        import vaex_amd
        sa = vaex_amd.superagg
        if not hasattr(sa, "code_column"):
            raise KeyError("no device coding in this library")
        kinds = {"int8": torch.int8, "int16": torch.int16, "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8, "float32": torch.float32, "float64": torch.float64}
        kinds.update({name: getattr(torch, name) for name in ("uint16", "uint32") if hasattr(torch, name)})   # (storage only: the library reads the bytes)
        device = int(sa.config_get("device"))
        free, _ = torch.cuda.mem_get_info(device)
        total = 0
        for name, c in columns.items():
            arrays = [c.data, c.mask] if isinstance(c, _Lazy) else [c]
            for a in arrays:
                if a is None:
                    continue
                if not isinstance(a, np.ndarray) or np.ma.isMaskedArray(a) or not a.dtype.isnative or (a.dtype.name not in kinds and not (isinstance(c, _Lazy) and a.dtype.kind == "b")):
                    raise KeyError(f"column {name!r} does not upload as it is")
                total += a.nbytes + (8 * len(a) if isinstance(c, _Lazy) and a is c.data else 0)
        if total * 3 >= free:
            raise MemoryError("the call's columns do not fit the device next to the partition queues")

        def upload(a):
            a = np.ascontiguousarray(a.view("u1") if a.dtype.kind == "b" else a)
            t = torch.empty(a.shape, dtype=kinds[a.dtype.name], device=f"cuda:{device}")
            sa.upload(a, t, 6)
            return t
        made, owners = {}, []
        for name, c in columns.items():
            if isinstance(c, _Lazy):
                d, m = upload(c.data), (None if c.mask is None else upload(c.mask))
                made[name], owner = c.device(sa, torch, d, m)
                owners.append(owner)
                del d, m
            else:
                made[name] = upload(c)
        stats["coded_on_device"] = stats.get("coded_on_device", 0) + len(lazies)
        return made, owners
    except (ImportError, KeyError, RuntimeError, MemoryError):
        stats["coded_on_host"] = stats.get("coded_on_host", 0) + len(lazies)
        return {name: (c.host() if isinstance(c, _Lazy) else c) for name, c in columns.items()}, []


def _frame_for(df, columns):
    from . import _cached_arrays
    columns, owners = _made_columns(columns)
    if owners:   # (everything sits in HBM already)
        from . import vaex_dist
        frame = binned.Frame(dict(columns), comm=vaex_dist.comm())
        frame._device_owners = owners
        return frame
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


def _block_as_numpy(block, offset=0):
    """a chunk as the executor hands it over -> a plain numpy array; missing values are outside the device groupby (RuntimeError: the task
    then leaves the pass to the others and vaex's own groupby answers afterwards).  A dictionary-encoded chunk (a categorical key of a
    _future() frame or an arrow dictionary column: vaex/dataframe.py:5835-5854) becomes its codes: the indices + the category offset."""
    if hasattr(block, "null_count") and hasattr(block, "type"):
        import pyarrow as pa
        if pa.types.is_dictionary(block.type):
            if block.null_count:
                raise RuntimeError("delayed groupby: a chunk with missing values")
            block = pa.concat_arrays([c.indices for c in block.chunks]) if isinstance(block, pa.ChunkedArray) else block.indices
            return np.asarray(block.to_numpy(zero_copy_only=False)) + offset if offset else np.asarray(block.to_numpy(zero_copy_only=False))
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
    if not 1 <= len(by_list) <= 8:
        return False
    try:
        for b in by_list:
            if isinstance(b, vaex.groupby.BinnerBase):   # (round 6: the object kinds _binner_object_key takes; the plan looks closer when the aggregation is known)
                if type(b) not in (vaex.groupby.Grouper, vaex.groupby.BinnerInteger, vaex.groupby.GrouperCategory, vaex.groupby.BinnerTime):
                    return False
                continue
            name = str(vaex.utils._ensure_string_from_expression(b))
            ar = df.columns.get(name)
            if ar is not None and (np.ma.isMaskedArray(ar) or (hasattr(ar, "null_count") and hasattr(ar, "type"))):
                # (a column with missing values / an arrow column: split or streamed when the aggregation is known — a cheap look at the type only)
                import pyarrow as pa
                if np.ma.isMaskedArray(ar):
                    ok = ar.dtype.kind in "biuf"
                else:
                    t = ar.type.index_type if pa.types.is_dictionary(ar.type) else ar.type
                    ok = pa.types.is_integer(t) or pa.types.is_floating(t) or pa.types.is_boolean(t)
                if not ok:
                    return False
                continue
            _real_column(df, name, _KEY_KINDS + _FLOAT_KEYS, "group key", materialise=False)   # (a cheap look: nothing is evaluated here)
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
                        [[name, repr(a)] for name, a in plan.actions if name in plan.finishers], [[name, nu["column"], nu["dropmissing"]] for name, nu in plan.nunique.items()],
                        # (round 6: how every key's codes were made and are decoded — two BinnerTime objects over one column differ only here)
                        [[name, m.get("kind"), m.get("what"), m.get("sort"), m.get("ascending"), m.get("dense"), m.get("null_code"),
                          vaex.cache.fingerprint(np.asarray(m["in_order"]).tolist() if "in_order" in m else None), vaex.cache.fingerprint(np.asarray(m["rank"]).tolist() if "rank" in m else None),
                          str(m.get("bins"))] for name, m in plan.key_meta.items()]]
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
                metas = self.plan.key_meta
                self.collector.append({name: _block_as_numpy(b, metas[name].get("offset", 0) if name in metas else 0) for name, b in zip(self.plan.columns, blocks)})
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
