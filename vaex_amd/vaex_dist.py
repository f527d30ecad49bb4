"""Row-sharded vaex behind install(): one process per GPU, every rank runs the SAME vaex program on its own rows.

vaex spreads a pass over the threads of one process and merges what they produced in `TaskPart.reduce`, called once per task by the
executor right before `get_result` (vaex/execution.py:451-453; TaskPartAggregation.reduce, vaex/cpu.py:788-796: Aggregator::merge per
grid; TaskPartStatistic.reduce, vaex/cpu.py:613-616: the statistic's own reduce function over the threads' grids).  An 8-GPU node is the
same picture one level up — ranks instead of threads — so this module puts the cross-rank merge in the same place:

    torch.distributed.init_process_group("nccl")          # one process per GPU (torchrun), RCCL over xGMI
    vaex_amd.install(distributed=True)
    df = vaex_amd.shard(vaex.open(...))                   # this rank's contiguous row range (vaex_amd.dist.shard_rows)
    df.mean("v", binby=["x", "y"], limits="minmax", shape=256)   # the whole table's answer, on every rank

* aggregation tasks (count / sum / moments / min / max, any binners, selections): after the local reduce every primitive grid is
  all-reduced in place (vaex_amd.dist.allreduce_aggs: the library's RCCL communicator on the device grids of the HIP classes; the
  (grids, ...) host buffers for task parts that were built from vaex's own C++ because install() does not offer the class);
* legacy statistic tasks (df.minmax, limits=None / "minmax", vaex/cpu.py:488-623): the ranks' grids are gathered and folded with
  the task's own `op.reduce` — the function vaex folds its threads' grids with;
* map_reduce tasks: the ranks' partial values are gathered and folded with the task's own reduce function;
* df.groupby(<integer keys>, agg=...) on the device groupby: the Frame behind it gets the communicator (key range / key union /
  partial groups agreed over the ranks: vaex_amd/binned.py).

Everything else that merges state vaex has no cross-process form for — distinct-key hash maps (vaex's own groupby passes, unique,
nunique), value_counts task parts, AggFirst / AggList / AggNUnique grids — raises NotImplementedError at its reduce instead of handing
back one rank's answer as if it were the table's.  Row counts (`len(df)`) stay the shard's.

Measured: world size 2 over gloo here (tests/test_vaex_dist_gloo.py: real vaex, shards vs the whole table), two HIP processes on one
GPU (-m gpu).  No multi-GPU node has run it: unmeasured on hardware, like everything in DESIGN section 4."""
import numpy as np

from . import dist as vdist

#: cross-rank merges performed, by task kind
stats = {"aggregations": 0, "statistics": 0, "map_reduce": 0}
_state = {"group": None, "on": False}


def active():
    """True when install(distributed=True) is in force and the process group has more than one rank"""
    if not _state["on"]:
        return False
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_world_size(_state["group"]) > 1


def comm():
    """the communicator the device groupby's Frame gets (None: single rank)"""
    return vdist.Comm(_state["group"]) if active() else None


def shard(df, group=None):
    """this rank's contiguous row range of `df` (a slice: no copy) — rows are independent"""
    import torch.distributed as dist
    if not dist.is_initialized():
        return df
    world = dist.get_world_size(group)
    if len(df) < world:
        # (a rank holding df[i:i] schedules no task part, never reaches reduce(), and the others wait in the all-reduce for ever)
        raise ValueError(f"vaex_amd.shard: {len(df)} rows over {world} ranks — every rank needs at least one row (a rank without rows would never reach the cross-rank merge)")
    i1, i2 = vdist.shard_rows(len(df), dist.get_rank(group), world)
    return df[i1:i2]


def _gather_objects(obj, group):
    import torch.distributed as dist
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


_NO_MERGE = ("AggFirst", "AggList", "AggNUnique")


def _aggregators_of(part):
    aggs = []
    for agg_desc, selections, aggregation, selection_waslist in part.aggregations:
        for si in range(len(selections)):
            a = aggregation[si]
            name = type(a).__name__
            if name.startswith(_NO_MERGE) or getattr(agg_desc, "name", "") in _NO_MERGE:
                raise NotImplementedError(f"row-sharded vaex: {name} has no cross-rank merge behind install() (vaex_amd.binned.Frame(comm=) has: first / last / nunique / list)")
            aggs.append(a)
    return aggs


def install(vaex_module, state, base_agg=None, group=None):
    """register task-part classes whose reduce() also merges across the ranks of `group`.  base_agg: the aggregation task part in
    force (install()'s HIP class; vaex's own in the CPU tests)."""
    import vaex.cpu
    _state["group"] = group
    _state["on"] = True
    base_agg = base_agg or vaex.cpu.TaskPartAggregation

    class TaskPartAggregationRanks(base_agg):
        snake_name = "aggregations"

        def reduce(self, others):
            base_agg.reduce(self, others)
            self._ranks_pending = False
            if active():
                aggs = _aggregators_of(self)
                if all(hasattr(a, "device_touch") for a in aggs):   # the HIP classes: in place, on the device grids where the backend is RCCL
                    vdist.allreduce_aggs(aggs, _state["group"])
                else:
                    # task parts built from vaex's own C++ (a class install() does not offer): their results are merged as host arrays
                    # in get_result — the reference's (grids, ...) buffer cannot be written through for grids of more than one
                    # dimension (its grid stride is wrong there: src/agg_base.hpp:115)
                    self._ranks_pending = True
                stats["aggregations"] += 1

        def get_result(self):
            results = base_agg.get_result(self)
            if getattr(self, "_ranks_pending", False):
                ops = []
                for agg_desc, selections, aggregation, selection_waslist in self.aggregations:
                    ops.append(vdist.agg_reduce_op(aggregation[0]))
                views = [np.ascontiguousarray(r).view(np.int64) if np.asarray(r).dtype.kind in "mM" else np.ascontiguousarray(r) for r in results]
                merged = vdist.allreduce_results(views, ops, _state["group"])
                results = [m.view(np.asarray(r).dtype).reshape(np.asarray(r).shape) for m, r in zip(merged, results)]
                self._ranks_pending = False
            return results

    base_stat = vaex.cpu.TaskPartStatistic

    class TaskPartStatisticRanks(base_stat):
        snake_name = "legacy_statistic"

        def reduce(self, others):
            base_stat.reduce(self, others)
            if active():
                grids = _gather_objects(np.asarray(self.grid), _state["group"])
                self.grid = self.op.reduce(np.array(grids))   # (vaex/cpu.py:613-616: the same fold, ranks instead of threads)
                stats["statistics"] += 1

    base_mr = vaex.cpu.TaskPartMapReduce

    class TaskPartMapReduceRanks(base_mr):
        snake_name = "map_reduce"

        def reduce(self, others):
            base_mr.reduce(self, others)
            if active():
                from functools import reduce as fold
                # (after the local reduce `values` is ONE folded value, or still the empty list of a rank without rows; maps that work by
                #  side effect — evaluate into a preallocated array — fold Nones: left alone)
                mine = self.values
                has = not (isinstance(mine, list) and len(mine) == 0) and mine is not None
                parts = [v for ok, v in _gather_objects((has, mine if has else None), _state["group"]) if ok]
                if parts:
                    self.values = fold(self._reduce, parts)
                    stats["map_reduce"] += 1

    def refusing(base, what):
        class Refuses(base):
            snake_name = base.snake_name

            def reduce(self, others):
                if active():
                    raise NotImplementedError(f"row-sharded vaex: {what} has no cross-rank merge behind install(); integer-key groupbys go through the device groupby, "
                                              "everything else of this kind through vaex_amd.binned.Frame(comm=)")
                return base.reduce(self, others)
        Refuses.__name__ = base.__name__ + "Ranks"
        return Refuses

    saved = {}
    for cls in (TaskPartAggregationRanks, TaskPartStatisticRanks, TaskPartMapReduceRanks,
                refusing(state.get("hash_task_hip") or vaex.cpu.TaskPartHashmapUniqueCreate, "a distinct-key hash map (vaex's own groupby / unique / nunique)"),
                refusing(vaex.cpu.TaskPartValueCounts, "value_counts")):
        vaex.cpu.register(cls)
        saved[cls.snake_name] = cls
    state["dist"] = dict(base_agg=base_agg, base_stat=base_stat, base_mr=base_mr, classes=saved,
                         base_hash=state.get("hash_task_hip") or vaex.cpu.TaskPartHashmapUniqueCreate, base_vc=vaex.cpu.TaskPartValueCounts)
    return saved


def uninstall(vaex_module, state):
    import vaex.cpu
    d = state.pop("dist", None)
    _state["on"] = False
    _state["group"] = None
    if d:
        for cls in (d["base_agg"], d["base_stat"], d["base_mr"], d["base_hash"], d["base_vc"]):
            vaex.cpu.register(cls)
