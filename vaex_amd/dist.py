"""Multi-GPU reduce of aggregator grids: rows are sharded over ranks (one process per GPU), every
rank bins its shard into private grids, and the per-rank grids are combined with ONE RCCL
all-reduce per grid over xGMI — the cross-rank form of Aggregator::merge
(/root/reference/packages/vaex-core/src/agg_count.cpp:15-23, agg_sum.cpp:72-79,
agg_minmax.cpp:19-26) driven by TaskPartAggregation.reduce (vaex/cpu.py:788-796).

Messages are tiny (259x259 cells x 8 B = 0.5 MB per grid) against 7 x ~153 GB/s xGMI links, so the
all-reduce is latency-bound; int64 counts reduce exactly, fp64 sums in the ring's fixed order.
The backend is whatever torch.distributed was initialised with: "nccl" (= RCCL) on GPUs, "gloo" in
the CPU tests (which exercise the same code path on host copies of the grids).
"""
import numpy as np

_KIND_OP = {0: "sum", 1: "sum", 2: "sum", 3: "min", 4: "max"}
_CLASS_KIND = {"AggCount_": 0, "AggSum_": 1, "AggSumMoment_": 2, "AggMin_": 3, "AggMax_": 4}


def agg_reduce_op(agg):
    """'sum' | 'min' | 'max' for an aggregator object of the superagg surface (by class name)."""
    name = type(agg).__name__
    for prefix, kind in sorted(_CLASS_KIND.items(), key=lambda kv: -len(kv[0])):
        if name.startswith(prefix):
            return _KIND_OP[kind]
    raise TypeError(f"not an aggregator: {name}")


def _reduce_op(name):
    import torch.distributed as dist
    return {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[name]


def allreduce_aggs(aggs, group=None, force=False):
    """In-place all-reduce of the grids of `aggs` across the process group.

    "nccl" backend (RCCL): the device grids are reduced in place over xGMI.  Any other backend ("gloo" in
    the CPU tests): the grids go through the aggregators' host buffers.  After the call every rank's
    aggregators hold the global result (get_result() returns it).

    Default on GPUs: the library's own RCCL communicator (`native_comm`, vxh_allreduce) — everything below about streams
    describes the torch path that VAEX_AMD_TORCH_ALLREDUCE=1 selects.
    Stream hand-off there (the grids belong to the library's streams, RCCL runs on torch's): asking an aggregator for its
    `__cuda_array_interface__` is vxh_agg_device_grid — it DRAINS the device (hipDeviceSynchronize), folds the replicas on
    the library's stream and waits for that fold before it returns (vxh_api.hip agg_fold_device), so the pointer RCCL gets
    is final and nothing of the library is in flight.  On the way back every all-reduce is waited for and the device
    synchronised before `device_touch()` hands the grid back to the library.  Both hand-offs are host-side full stops; the
    grids are 0.5 - 18 MB, once per pass."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return
    if dist.get_backend(group) != "nccl" or not all(hasattr(a, "device_touch") for a in aggs):
        return allreduce_aggs_host(aggs, group)
    native = native_comm(group)
    if native is not None:
        # the library's own collective (vxh_allreduce, RCCL directly): fold + all-reduce on ITS stream, stream-ordered with the
        # binning before and whatever follows — no hand-off to torch's stream, no host-side stop
        native.allreduce(list(aggs))
        return
    works = []
    tensors = []
    for agg in aggs:
        op = agg_reduce_op(agg)
        iface = agg.__cuda_array_interface__  # folds the replicas; device pointer of the result grid
        t = torch.as_tensor(_Wrap(iface, op), device="cuda")
        tensors.append(t)
        works.append(dist.all_reduce(t, op=_reduce_op(op), group=group, async_op=True))
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    for agg in aggs:
        agg.device_touch()


_NATIVE = {}


def native_comm(group=None):
    """the library's RCCL communicator (vaex_amd.superagg.Comm = vxh_comm_init) over the ranks of `group`, created on first
    use: rank 0's unique id travels through torch.distributed's object broadcast (any transport would do — a non-Python host
    of libvaexhip.so uses its own).  None when the extension has no Comm or VAEX_AMD_TORCH_ALLREDUCE=1 asks for torch's
    collectives (the round-3 path, kept for A/B runs)."""
    import os
    import torch.distributed as dist
    if os.environ.get("VAEX_AMD_TORCH_ALLREDUCE") == "1":
        return None
    from . import superagg as sa
    if not hasattr(sa, "Comm"):
        return None
    key = id(group) if group is not None else 0
    if key not in _NATIVE:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [sa.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        _NATIVE[key] = sa.Comm(world, rank, box[0])
    return _NATIVE[key]


def allreduce_aggs_host(aggs, group=None, reduce_arrays=None):
    """Same reduce through host copies of the results.  reduce_arrays(arrays, ops) -> arrays: the exchange itself (default:
    allreduce_results over `group`).

    The HIP classes take the global result back through their (grids, *shapes) host buffer (buffer protocol of the superagg
    surface, src/agg_base.hpp:106-125): grid 0 receives it, the others the identity, and get_result() folds every replica.
    Aggregators of any OTHER class (the reference's own C++ in the CPU tests) are not written to — their get_result() folds
    only the grids whose thread slot saw a row and RE-INITIALISES grid 0 when slot 0 never did (`if (!grid_used[0])
    initial_fill(0)`, src/agg_count.cpp:24-41; which slots see rows depends on how the pool's threads were scheduled, and a
    rank without rows uses none), and their buffer's grid stride is wrong beyond one dimension (src/agg_base.hpp:115).  For
    those the reduced arrays are RETURNED, one per aggregator (None where the aggregator was written through), and the
    caller reads them instead of get_result() — vaex_amd.binned.Frame does, vaex_dist merges results the same way."""
    ops = [agg_reduce_op(a) for a in aggs]
    local = [np.array(a.get_result()) for a in aggs]
    reduced = reduce_arrays(local, ops) if reduce_arrays is not None else allreduce_results(local, ops, group)
    returned = []
    for a, r, op in zip(aggs, reduced, ops):
        if not hasattr(a, "device_touch"):
            returned.append(np.asarray(r))
            continue
        returned.append(None)
        buf = np.asarray(a)
        buf[0] = r
        if buf.shape[0] > 1:
            ident = 0
            if op != "sum":
                info = np.finfo(buf.dtype) if buf.dtype.kind == "f" else (np.iinfo(buf.dtype) if buf.dtype.kind in "iu" else None)
                if buf.dtype.kind == "f":
                    ident = np.inf if op == "min" else -np.inf
                elif info is not None:
                    ident = info.max if op == "min" else info.min
                else:
                    ident = op == "min"
            buf[1:] = ident
    return returned if any(r is not None for r in returned) else None


class ReducedAgg:
    """Stand-in for an aggregator that could not take the cross-rank result back (see allreduce_aggs_host): get_result() is
    the reduced array, everything else is the aggregator's."""

    def __init__(self, agg, result):
        self._agg, self._result = agg, result

    def get_result(self):
        return self._result

    def __getattr__(self, name):
        return getattr(self._agg, name)


def with_reduced(aggs, returned):
    """aggs with the ones `reduce(aggs)` returned arrays for replaced by ReducedAgg stand-ins"""
    if returned is None:
        return aggs
    return [a if r is None else ReducedAgg(a, r) for a, r in zip(aggs, returned)]


class _Wrap:
    """Adapter so torch can alias the library-owned grid: uint64 cells are presented as int64 (the sum
    of the two's-complement bit patterns is the same)."""

    def __init__(self, iface, op):
        iface = dict(iface)
        if iface["typestr"] == "<u8":
            if op != "sum":
                raise NotImplementedError("min/max all-reduce of uint64 grids: use allreduce_results()")
            iface["typestr"] = "<i8"
        self.__cuda_array_interface__ = iface


def allreduce_results(results, ops, group=None):
    """All-reduce HOST result arrays (list of ndarrays) — used by the gloo CPU tests and as the fallback
    for cell types RCCL cannot reduce.  Returns new arrays."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [np.array(r) for r in results]
    out = []
    for r, op in zip(results, ops):
        a = np.ascontiguousarray(r)
        view = a.view(np.int64) if a.dtype == np.uint64 else a
        t = torch.from_numpy(view.copy())
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=_reduce_op(op), group=group)
        res = t.cpu().numpy()
        out.append(res.view(np.uint64).reshape(r.shape) if a.dtype == np.uint64 else res.reshape(r.shape))
    return out


def shard_rows(n, rank, world):
    """Contiguous row range [i1, i2) of `rank` (rows are independent: SURVEY §8e)."""
    # balanced: the first n % world ranks take one row more — no rank is empty while n >= world (the ceil(n / world) split of rounds 2-4
    # left the last ranks without rows for small n; a rank without rows never reaches vaex's reduce: vaex_dist.shard)
    base, rem = divmod(n, world)
    i1 = rank * base + min(rank, rem)
    return i1, i1 + base + (1 if rank < rem else 0)


class Comm:
    """The two small exchanges a multi-rank groupby needs besides the grid all-reduce."""

    def __init__(self, group=None):
        self.group = group

    def _device(self):
        import torch.distributed as dist
        return "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"

    def minmax(self, lo, hi):
        """global (min, max) of per-rank integer ranges.  A rank without rows passes (INT64_MAX, INT64_MIN): the two
        ends are reduced separately (MIN of the lows, MAX of the highs) — negating INT64_MIN does not fit int64."""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return lo, hi
        tlo = torch.tensor([lo], dtype=torch.int64, device=self._device())
        thi = torch.tensor([hi], dtype=torch.int64, device=self._device())
        dist.all_reduce(tlo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(thi, op=dist.ReduceOp.MAX, group=self.group)
        return int(tlo[0]), int(thi[0])

    def sum_ints(self, values):
        """element-wise sum over the ranks of a short list of Python ints (row counts of NaN / missing values)"""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return [int(v) for v in values]
        t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=self._device())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.cpu().tolist()]

    def minmax_float(self, lo, hi):
        """global (min, max) of per-rank float ranges (df.minmax / limits=None under row sharding); a rank without rows
        contributes (+inf, -inf)"""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return lo, hi
        t = torch.tensor([lo, -hi], dtype=torch.float64, device=self._device())
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return float(t[0]), -float(t[1])

    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def rank(self):
        import torch.distributed as dist
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def all_gather_arrays(self, arrays):
        """every rank's list of 1-d numpy arrays (same dtypes on every rank, any lengths) -> [rank][array].  The bytes travel as
        tensors of the backend's device (RCCL over xGMI on GPUs: one all_gather of the lengths, one padded all_gather per
        array) — no pickling, no host round trip of the collective itself."""
        import torch
        import torch.distributed as dist
        arrays = [np.ascontiguousarray(a) for a in arrays]
        world = self.world()
        if world == 1:
            return [arrays]
        dev = self._device()
        lens = torch.tensor([a.nbytes for a in arrays], dtype=torch.int64, device=dev)
        all_lens = [torch.empty_like(lens) for _ in range(world)]
        dist.all_gather(all_lens, lens, group=self.group)
        all_lens = [t.cpu().tolist() for t in all_lens]
        out = [[] for _ in range(world)]
        for j, a in enumerate(arrays):
            longest = max(l[j] for l in all_lens)
            buf = torch.zeros(max(longest, 1), dtype=torch.uint8, device=dev)
            if a.nbytes:
                buf[:a.nbytes] = torch.from_numpy(a.reshape(-1).view(np.uint8)).to(dev)
            parts = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(parts, buf, group=self.group)
            for r in range(world):
                out[r].append(parts[r][:all_lens[r][j]].cpu().numpy().view(a.dtype).copy())
        return out

    def all_agree(self, ok):
        """True when `ok` holds on EVERY rank (one MIN all-reduce): ranks must take the same branch before a collective"""
        import torch
        import torch.distributed as dist
        if self.world() == 1:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=self._device())
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t[0]))

    def union_keys(self, keys):
        """sorted union of the ranks' distinct keys (<= 1e6 x 8 B per rank in the BASELINE config)"""
        if self.world() == 1:
            return keys
        parts = self.all_gather_arrays([np.asarray(keys)])
        return np.unique(np.concatenate([p[0] for p in parts]))

    def allreduce(self, aggs):
        return allreduce_aggs(aggs, self.group)

    def allreduce_arrays(self, arrays, ops):
        """element-wise 'sum' / 'min' / 'max' of host arrays over the ranks (tensors of the backend's device underneath)"""
        return allreduce_results(arrays, ops, self.group)
