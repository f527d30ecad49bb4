"""vaex_amd — MI355X-native N-d binned statistics / groupby aggregation for vaex.

The only thing in here is vaex's one data-parallel hot path (the `vaex.superagg` kernels behind
df.count/sum/mean/std(binby=...) and df.groupby().agg()), as hand-written HIP kernels in
libvaexhip.so (C-ABI: include/vaex_hip.h) behind the reference's own class surface:

    vaex_amd.superagg      pybind11 shim with the classes of `vaex.superagg`
    vaex_amd.hashset       `vaex.superutils.ordered_set_<dtype>` on the GPU hash map
    vaex_amd.install()     plug both into an (unmodified) vaex installation, task by task
    vaex_amd.vaexfast      the legacy `vaex.vaexfast.statisticNd_f8` entry on the same kernels
    vaex_amd.binned        host-side driver mirroring df.count/sum/mean/...(binby=, limits=, shape=)
    vaex_amd.dist          row-sharded multi-GPU reduce of the grids over RCCL

There is no CPU fallback inside the library: without a GPU every compute call raises.  What install() falls
back to for aggregations the HIP classes do not offer is vaex's OWN C++ — see `install`.
"""
# torch (when present) must be imported BEFORE libvaexhip.so is loaded: torch ships its own
# libamdhip64.so (SONAME libamdhip64.so.7); loading it first makes our library bind to the same
# HIP runtime instead of bringing a second one into the process.
try:  # pragma: no cover - plumbing
    import torch as _torch  # noqa: F401
except Exception:  # torch is only needed for device tensors / torch.distributed
    _torch = None

import threading as _threading

from . import superagg  # noqa: E402  (fails loudly if the extension was not built)

__all__ = ["superagg", "install", "uninstall", "cache_columns", "uncache_columns"]

# names of the reference's vaex.superagg that the HIP module deliberately does not take over (none of the binner / numeric aggregator
# classes any more: BinnerHash_<T>[_non_native] takes the reference's constructor and cells since round 3, vaex_amd/hashset.py +
# vxh_binner_hash_create_ref; vaex itself cannot build one — vaex/cpu.py:63 calls the three-argument constructor with two)
_HIDDEN_PREFIXES = ()
UNSUPPORTED = ("AggCount_string", "AggCount_object", "*_string / *_object aggregators", "BinnerCombined", "BinnerHash_string")


class _Backend:
    """What `vaex.superagg` is after install(): a per-thread switch between the HIP classes and vaex's own C++ module.

    vaex looks classes up BY NAME on this object while it builds a task part (vaex/utils.py:754-791 from
    vaex/cpu.py:44-65 and vaex/agg.py:278-321).  In "hip" mode a name the HIP module does not have raises AttributeError
    — find_type_from_dtype turns that into its ValueError — and the task part is then rebuilt in "cpu" mode, i.e. entirely
    from the reference's classes: a task is never a mix of the two (their grids are different objects)."""

    def __init__(self, hip, cpu):
        self.__dict__["_hip"] = hip
        self.__dict__["_cpu"] = cpu
        self.__dict__["_tls"] = _threading.local()
        self.__dict__["__name__"] = "vaex.superagg"

    def _mode(self):
        return getattr(self._tls, "mode", "hip")

    def use(self, mode):
        backend = self

        class _Ctx:
            def __enter__(self_):
                self_.prev = backend._mode()
                backend._tls.mode = mode

            def __exit__(self_, *exc):
                backend._tls.mode = self_.prev
                return False
        return _Ctx()

    def __getattr__(self, name):
        if self._mode() == "cpu":
            return getattr(self._cpu, name)
        if _HIDDEN_PREFIXES and name.startswith(_HIDDEN_PREFIXES):
            raise AttributeError(name)
        return getattr(self._hip, name)

    def __dir__(self):
        return dir(self._hip)


_installed = {}


#: where the aggregation task parts of an installed vaex were built (one per task and pool decode): on the HIP classes, or on vaex's own
#: C++ because an aggregator class / dtype is not offered here — with the reason, so that "drop-in" can be audited per workload
task_stats = {"hip": 0, "cpu": 0, "cpu_reasons": {}}

AUTO_CHUNK_ROWS_MAX = 1 << 26   # install(chunk_size="auto"): upper bracket of vaex's automatic chunk size (rows)


def install(vaex_module=None, legacy=True, hash_sets=True, chunk_size="auto", groupby=True, selections=True, filters=True, distributed=False, group=None):
    """Plug the HIP kernels into an unmodified vaex.

    * `vaex.superagg` becomes a `_Backend` and the task-part registry entry "aggregations" (vaex/cpu.py:629-631,
      vaex/encoding.py:31-52) a subclass of vaex's TaskPartAggregation whose `decode` builds the task part from the HIP
      classes and, when an aggregation or binner it needs is not among them (string and
      object aggregators, BinnerCombined: `vaex_amd.UNSUPPORTED`), builds it again from vaex's own C++ —
      so everything that worked before install() still works, on the CPU, and everything on the hot path runs on the GPU.
    * legacy=True points the legacy statistic task (df.minmax / limits=None: vaex/cpu.py:533-538) at
      vaex_amd.vaexfast.statisticNd_f8 / statisticNd_f4 (the float32 entry with the reference's float32 scaling arithmetic).
    * hash_sets=True replaces `vaex.hash.ordered_set_<dtype>` for the numeric dtypes (vaex/hash.py:49-52 looks them up
      by name) with vaex_amd.hashset's GPU-backed classes: groupby's distinct-key pass and `_ordinal_values`.
    * chunk_size: vaex's executor cuts a pass into chunks of `rows / threads` rows bracketed by [chunk.size_min, chunk.size_max
      = 1 Mi] (vaex/execution.py:283-292) — a bracket chosen for CPU caches: at 1 Mi rows a chunk is bound by its fixed costs
      here (the Python of TaskPartAggregation.process, one launch: profiles/r02_vaex_dropin_timing.txt, 8 Grows/s from HBM).
      "auto" (default) raises the UPPER bracket, vaex.settings.main.chunk.size_max, to AUTO_CHUNK_ROWS_MAX = 64 Mi rows unless
      the user changed it, so that every pool thread gets one chunk of rows / threads rows up to that size and nobody has to
      pass anything; an integer sets vaex.settings.main.chunk.size itself; None leaves vaex's chunking alone.
    * selections=True: selection expressions of the comparison subset (`column <op> number` joined by & | ~, <= 4 terms) are
      evaluated on the device instead of numpy (vaex_amd/vaex_selection.py) — given as an expression or as a NAMED selection whose
      history (df.select modes, select_inverse, undo / redo) resolves to one, over numpy columns or arrow columns without nulls;
      everything else keeps vaex's host masks.
    * filters=True (needs selections=True): binned aggregations over a FILTERED frame (df[df.x > 0]) get their chunks uncompacted and
      take the filter as a keep-mask — a device predicate where it compiles — instead of vaex copying every column through a boolean
      index per chunk (vaex_amd/vaex_filter.py); the device column cache then serves filtered frames too.
    * groupby=True: df.groupby(<integer key columns>, agg=count / sum / mean / var / std ...) is answered by the device
      groupby (vaex_amd/vaex_groupby.py) instead of vaex's two passes; everything else falls through to vaex's own code.
    * distributed=True (torch.distributed initialised, one process per GPU; `group`: the ranks that share the table): every rank
      runs the same vaex program on its shard (`vaex_amd.shard(df)`), and the task parts' reduce() — where vaex merges its threads
      (vaex/cpu.py:788-796) — also merges across the ranks: one RCCL all-reduce per aggregator grid (vaex_amd/vaex_dist.py)."""
    import sys
    if vaex_module is None:
        import vaex as vaex_module
    import vaex.cpu
    if "backend" in _installed:
        return _installed["backend"]
    cpu_module = vaex_module.superagg
    # the kernels' code objects and thread slot 0 now, not inside the user's first df.count / df.groupby (a fresh process paid 0.2-0.6 s
    # there: profiles/r06_process_first.txt); without a device install() still succeeds — every compute call then raises, as before
    try:
        if superagg.device_count() > 0:
            superagg.warmup()
    except RuntimeError:
        pass
    backend = _Backend(superagg, cpu_module)
    _installed.update(backend=backend, cpu_module=cpu_module, vaex=vaex_module, task_cls=vaex.cpu.TaskPartAggregation)
    vaex_module.superagg = backend
    sys.modules["vaex.superagg"] = backend

    base = vaex.cpu.TaskPartAggregation

    class TaskPartAggregationHip(base):
        snake_name = "aggregations"
        backend_used = "hip"

        @classmethod
        def decode(cls, encoding, spec, df, nthreads):
            from . import vaex_selection, vaex_filter
            import vaex.memory
            as_mask = bool(spec.get(vaex_filter.SPEC_KEY, False))   # a filtered frame's blocks arrive uncompacted (vaex_amd/vaex_filter.py)
            named = spec.get(vaex_selection.SPEC_KEY)                # what the task's named selections stood for when scheduled
            # the executor checks the parts' memory_usage() against what its tracker saw (vaex/execution.py:413-414): aggregators a
            # failed HIP attempt built before it hit an unsupported one must not stay on the tracker's books
            tracker = getattr(vaex.memory.local, "agg", None)
            booked = getattr(tracker, "used", None)
            with backend.use("hip"):
                try:
                    part = base.decode.__func__(cls, encoding, spec, df, nthreads)
                    part.backend_used = "hip"
                    vaex_selection.attach(part, "hip", superagg, nthreads, filter_as_mask=as_mask, named=named)
                    task_stats["hip"] += 1
                    return part
                except (ValueError, TypeError, NotImplementedError) as e:
                    if "Could not find a class" not in str(e) and not isinstance(e, NotImplementedError):
                        raise
                    task_stats["cpu_reasons"][str(e)[:120]] = task_stats["cpu_reasons"].get(str(e)[:120], 0) + 1
            if booked is not None:
                tracker.used = booked
            with backend.use("cpu"):
                part = base.decode.__func__(cls, encoding, spec, df, nthreads)
                part.backend_used = "cpu"
                vaex_selection.attach(part, "cpu", superagg, nthreads, filter_as_mask=as_mask, named=named)
                task_stats["cpu"] += 1
                return part

        def process(self, thread_index, i1, i2, filter_mask, selection_masks, blocks):
            # selections planned for the device (vaex_amd/vaex_selection.py): their columns' chunks ride behind the blocks
            from . import vaex_selection, vaex_filter
            selection_masks, blocks = vaex_selection.before_process(self, thread_index, selection_masks, blocks)
            if getattr(self, "_hip_filter_as_mask", False) and filter_mask is not None:
                return vaex_filter.process(self, base, thread_index, i1, i2, filter_mask, selection_masks, blocks)
            return base.process(self, thread_index, i1, i2, filter_mask, selection_masks, blocks)

    vaex.cpu.register(TaskPartAggregationHip)
    _installed["task_hip"] = TaskPartAggregationHip

    base_h = vaex.cpu.TaskPartHashmapUniqueCreate

    class TaskPartHashmapUniqueCreateHip(base_h):
        """the distinct-key pass with a GPU-backed set: the executor compares the parts' memory_usage() with what its
        tracker saw (vaex/execution.py:413-414) — sys.getsizeof of a Python object includes the GC header the
        reference's pybind type does not have, so the bytes are asked from the set itself"""
        snake_name = base_h.snake_name

        def memory_usage(self):
            internal = self.hash_map_unique._internal
            return internal.bytes_used() if hasattr(internal, "bytes_used") else sys.getsizeof(internal)

    _installed["hash_task_cls"] = base_h
    _installed["hash_task_hip"] = TaskPartHashmapUniqueCreateHip
    vaex.cpu.register(TaskPartHashmapUniqueCreateHip)
    if distributed:
        from . import vaex_dist
        vaex_dist.install(vaex_module, _installed, base_agg=TaskPartAggregationHip, group=group)
    if legacy:
        from . import vaexfast as _vf
        legacy_mod = getattr(vaex_module, "vaexfast", None) or sys.modules.get("vaex.vaexfast")
        if legacy_mod is not None:
            _installed["legacy"] = (legacy_mod, legacy_mod.statisticNd_f8)
            legacy_mod.statisticNd_f8 = _vf.statisticNd_f8
            if hasattr(legacy_mod, "statisticNd_f4"):
                original_f4 = legacy_mod.statisticNd_f4
                _installed["legacy_f4"] = original_f4

                def statisticNd_f4(*args, **kwargs):   # (OP_COV over float32 weights stays with the reference's function)
                    try:
                        return _vf.statisticNd_f4(*args, **kwargs)
                    except NotImplementedError:
                        return original_f4(*args, **kwargs)
                legacy_mod.statisticNd_f4 = statisticNd_f4
    if chunk_size == "auto":
        if vaex_module.settings.main.chunk.size_max == 1024 ** 2:  # (vaex's default: the user has not chosen one)
            _installed["chunk_size_max"] = vaex_module.settings.main.chunk.size_max
            vaex_module.settings.main.chunk.size_max = AUTO_CHUNK_ROWS_MAX
    elif chunk_size is not None:
        _installed["chunk_size"] = vaex_module.settings.main.chunk.size
        vaex_module.settings.main.chunk.size = int(chunk_size)
    if groupby:
        from . import vaex_groupby
        vaex_groupby.install(vaex_module, _installed)
    if selections:
        from . import vaex_selection
        vaex_selection.install(vaex_module, _installed)
        if filters:
            from . import vaex_filter
            vaex_filter.install(vaex_module, _installed)
    if hash_sets:
        import copyreg
        import vaex.hash
        from . import hashset
        saved = {}
        for name, cls in hashset.CLASSES.items():
            attr = "ordered_set_" + name
            if hasattr(vaex.hash, attr):
                saved[attr] = getattr(vaex.hash, attr)
                setattr(vaex.hash, attr, cls)
                copyreg.pickle(cls, lambda x: x.__reduce__())
        _installed["hash"] = saved
        _installed["hash_tuple"] = vaex.hash.ordered_set
        vaex.hash.ordered_set = tuple(vaex.hash.ordered_set) + tuple(hashset.CLASSES.values())
    return backend


def shard(df, group=None):
    """this rank's contiguous row range of a vaex DataFrame (install(distributed=True): vaex_amd/vaex_dist.py)"""
    from . import vaex_dist
    return vaex_dist.shard(df, group)


def uninstall():
    """undo install() (tests)"""
    import sys
    if "backend" not in _installed:
        return
    import vaex.cpu
    import vaex.hash
    vaex_module = _installed["vaex"]
    vaex_module.superagg = _installed["cpu_module"]
    sys.modules["vaex.superagg"] = _installed["cpu_module"]
    if "dist" in _installed:
        from . import vaex_dist
        vaex_dist.uninstall(vaex_module, _installed)
    vaex.cpu.register(_installed["task_cls"])
    if "hash_task_cls" in _installed:
        vaex.cpu.register(_installed["hash_task_cls"])
    if "legacy" in _installed:
        mod, fn = _installed["legacy"]
        mod.statisticNd_f8 = fn
        if "legacy_f4" in _installed:
            mod.statisticNd_f4 = _installed["legacy_f4"]
    for attr, cls in _installed.get("hash", {}).items():
        setattr(vaex.hash, attr, cls)
    if "hash_tuple" in _installed:
        vaex.hash.ordered_set = _installed["hash_tuple"]
    if "chunk_size" in _installed:
        vaex_module.settings.main.chunk.size = _installed["chunk_size"]
    if "chunk_size_max" in _installed:
        vaex_module.settings.main.chunk.size_max = _installed["chunk_size_max"]
    if "groupby" in _installed:
        from . import vaex_groupby
        vaex_groupby.uninstall(vaex_module, _installed)
    if "filter" in _installed:
        from . import vaex_filter
        vaex_filter.uninstall(vaex_module, _installed)
    if "selection" in _installed:
        from . import vaex_selection
        vaex_selection.uninstall(vaex_module, _installed)
    _installed.clear()


# ---------------------------------------------------------------------------------------------------------------------
# device column cache (include/vaex_hip.h "chunk feeder and device column cache")
# ---------------------------------------------------------------------------------------------------------------------
_cached_arrays = {}


def _host_arrays(obj, columns=None):
    """the numpy arrays behind `obj`: a vaex DataFrame (df.columns: memory-mapped / numpy columns, the value buffers of null-free
    arrow columns; virtual columns are skipped), a dict of arrays, or one array"""
    import numpy as np
    if hasattr(obj, "columns") and hasattr(obj, "get_column_names"):
        names = columns if columns is not None else list(obj.columns)
        items = [(n, obj.columns[n]) for n in names if n in obj.columns]
    elif isinstance(obj, dict):
        items = [(n, a) for n, a in obj.items() if columns is None or n in columns]
    else:
        items = [("array", obj)]
    out = []
    for name, a in items:
        if np.ma.isMaskedArray(a):
            parts = [np.ma.getdata(a)] + ([np.ma.getmaskarray(a)] if a.mask is not np.ma.nomask else [])
        elif isinstance(a, np.ndarray):
            parts = [a]
        else:
            # arrow-backed columns (pyarrow.ChunkedArray: what vaex holds for arrow / parquet files, vaex/arrow/dataset.py:164-200):
            # the value buffers of null-free chunks of a fixed-width numeric type, as zero-copy numpy views (bool is bit-packed
            # and chunks with nulls carry a validity bitmap: those keep streaming)
            parts = _arrow_value_buffers(a)
        for part in parts:
            if part.ndim == 1 and part.flags.c_contiguous and part.dtype.kind in "iufb" and part.nbytes:
                out.append(part)
    return out


def _arrow_value_buffers(column):
    try:
        import pyarrow as pa
    except ImportError:
        return []
    import numpy as np
    if isinstance(column, pa.Array):
        chunks = [column]
    elif isinstance(column, pa.ChunkedArray):
        chunks = column.chunks
    else:
        return []
    out = []
    for chunk in chunks:
        t = chunk.type
        if chunk.null_count or not (pa.types.is_integer(t) or pa.types.is_floating(t)) or len(chunk) == 0:
            continue
        try:
            out.append(chunk.to_numpy(zero_copy_only=True))   # (a view: keeps the arrow buffer alive)
        except (pa.ArrowInvalid, ValueError):
            continue
    return out


def cache_columns(obj, columns=None, pin=True):
    """Declare the columns' memory immutable and let their chunks stay in HBM between passes: the first
    df.count/mean/...(binby=) over them streams the chunks across PCIe (straight from the page-locked columns when
    `pin`), every later pass finds them on the device.  Returns the number of bytes registered.  The arrays are kept
    alive until uncache_columns(); writing to them while registered gives stale results (vaex columns are immutable)."""
    total = 0
    for a in _host_arrays(obj, columns):
        key = a.__array_interface__["data"][0]
        if key in _cached_arrays:
            continue
        superagg.cache_register(a.view("u1") if a.dtype.kind == "b" else a, pin)
        _cached_arrays[key] = a
        total += a.nbytes
    return total


def uncache_columns(obj=None, columns=None):
    """forget the columns registered by cache_columns (all of them when obj is None) and free their device chunks"""
    arrays = list(_cached_arrays.values()) if obj is None else _host_arrays(obj, columns)
    for a in arrays:
        key = a.__array_interface__["data"][0]
        if key in _cached_arrays:
            superagg.cache_unregister(a.view("u1") if a.dtype.kind == "b" else a)
            del _cached_arrays[key]
            import sys
            vg = sys.modules.get(__name__ + ".vaex_groupby")
            if vg is not None:   # (the wrapped df.groupby keeps device copies of registered columns)
                vg._device_copies.pop(key, None)
