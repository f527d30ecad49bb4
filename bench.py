#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: rows/s of the 2-D count+mean pass on a 256x256 grid,
1e9 float64 rows per GPU, HBM-resident, N GPUs of one node (one process per GPU, RCCL).

A "step" = one full pass over this rank's rows: reset grids, ONE fused kernel pass computing
count(*), sum(v), count(v) (= what df.count + df.mean(v) binby=[x,y] shape=256 schedules:
vaex/agg.py:391-418 builds mean from sum and count, merged into one TaskAggregations pass),
fold of the replicas, (N>1) RCCL all-reduce of the three grids, D2H of the results and the numpy
finish mean = sum/count.  Inputs are synthetic N(0,1)/N(3,2) columns generated on the device
(there is no dataset on the box); limits [-4,4].

`--gpus N` with N > 1 and no torchrun environment (WORLD_SIZE unset) makes this process spawn the N ranks itself
(torch.multiprocessing, one process per GPU, RCCL); under `torch.distributed.run` the ranks come from the environment.
Rows per GPU (weak scaling): 1e9 at N = 1 — BASELINE configs[1], the BENCH line — and 1.25e9 at N > 1, the shard of configs[4]
(1e10 rows row-sharded over 8 GPUs: `--gpus 8` runs exactly that config; 2 and 4 GPUs the same shard).  `--total-rows T` splits ONE
table over the ranks instead (strong scaling).  At N > 1 the line also carries north_star's target sentence (2-D count(*), 16 B/row)
on the same shards as `configs[0]`.  `--backend gloo` sends the grids through host buffers, so the ranks may share a GPU: the dry run
of launcher, sharding, reduce and JSON on a one-GPU box (tests/test_bench_contract.py).

Besides `value` (N(0,1) data, the hot box warm) the N=1 line carries the two unflattering numbers of the same pass:
`value_uniform` (x,y ~ U(-4,4): the densest box holds only ~15 % of the rows, the rest goes through the partition queues) and `value_cold` (the hot box
sampled anew in every step, as a first call on fresh columns pays it).

Prints ONE JSON line (rank 0) with `roofline` (HIP-event kernel time on the library's stream vs
24 B/row of compulsory reads) and `cpu_baseline` (the reference's own C++ from oracle/_ref driven
from a thread pool like vaex's executor, or the C port if that is absent) objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
BYTES_PER_ROW = 24     # x, y, v float64: compulsory reads of the 2-D count+mean pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=float, default=None, help="rows per GPU (weak scaling).  Default: 1e9 at N = 1 (BASELINE configs[1]); 1.25e9 at N > 1 — the "
                    "shard of configs[4] (1e10 rows row-sharded over 8 GPUs: N = 8 runs exactly that config, N = 2 / 4 the same shard on fewer GPUs)")
    ap.add_argument("--total-rows", type=float, default=None, help="rows of the WHOLE job, split evenly over the ranks (strong scaling; overrides --rows)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend at N > 1: nccl (= RCCL over xGMI, one GPU per rank) or gloo "
                    "(the grids cross through host buffers; the ranks may then SHARE a GPU — the N = 2 dry run of launcher, sharding, reduce and JSON on a one-GPU box)")
    ap.add_argument("--shape", type=int, default=256)
    ap.add_argument("--cpu-rows", type=float, default=1e8, help="rows of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip value_uniform / value_cold")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` section (BASELINE configs[2] / configs[3] on the same clock)")
    return ap.parse_args()


_VAEX_CPU_SCRIPT = r"""
import json, os, sys, time
import numpy as np
pkg, fake, tmp, shape = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
sys.path[:0] = [pkg, fake]
import vaex
x, y, v = (np.load(os.path.join(tmp, c + ".npy"), mmap_mode="r") for c in "xyv")
df = vaex.from_arrays(x=x, y=y, v=v)
lim = [[-4.0, 4.0], [-4.0, 4.0]]
best, c, m = 1e30, None, None
for rep in range(4):
    t0 = time.perf_counter()
    c = df.count(binby=["x", "y"], limits=lim, shape=shape, delay=True)
    m = df.mean("v", binby=["x", "y"], limits=lim, shape=shape, delay=True)
    df.execute()                      # ONE pass of vaex's executor: count(*), sum(v), count(v) fused (vaex/cpu.py:678-786)
    dt = time.perf_counter() - t0
    if rep:
        best = min(best, dt)
c, m = np.asarray(c.get()), np.asarray(m.get())
print(json.dumps({"s": best, "rows": int(len(df)), "threads": int(vaex.settings.main.thread_count), "counted": int(c.sum()), "mean_finite_cells": int(np.isfinite(m).sum())}))
"""


def vaex_cpu_pass(xs, ys, vs, shape, cores):
    """north_star's "vaex's own multithreaded CPU path": the SAME call through the real vaex package (the reference's Python from
    oracle/_ref/vaexpy over the reference's C++ from oracle/_ref, VAEX_NUM_THREADS = all host cores, vaex's own executor, chunking and
    expression layer) in a subprocess, on the same sample of rows.  None when the package is not on this box."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(root, "oracle", "_ref", "vaexpy")
    if not os.path.isdir(os.path.join(pkg, "vaex")):
        pkg = os.path.join(root, "oracle", "_ref", "overlay")
    if not os.path.isdir(os.path.join(pkg, "vaex")):
        return None
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for name, a in zip("xyv", (xs, ys, vs)):
            np.save(os.path.join(tmp, name + ".npy"), a)
        env = dict(os.environ, VAEX_NUM_THREADS=str(cores), VAEX_HOME=tmp)
        out = subprocess.run([sys.executable, "-c", _VAEX_CPU_SCRIPT, pkg, os.path.join(root, "oracle", "fake"), tmp, str(shape)], cwd="/tmp", env=env,
                             capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            return {"error": (out.stderr or out.stdout)[-400:]}
        r = json.loads(out.stdout.strip().splitlines()[-1])
        return {"value": r["rows"] / r["s"], "unit": "rows/s", "threads": r["threads"], "rows_inside_the_limits": [r["counted"], r["rows"]],
                "call": "df.count(binby=[x,y], delay=True) + df.mean(v, binby=[x,y], delay=True) + df.execute(): one executor pass, best of 3 after a warm-up"}
    except Exception as e:   # (the bench line must not depend on it)
        return {"error": repr(e)}
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(x, y, v, shape, rows):
    """The same pass on the host cores: reference C++ (oracle/_ref) when it loads, else the C port."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    rows = int(min(rows, len(x)))
    xs, ys, vs = (t[:rows].cpu().numpy() for t in (x, y, v))
    ref = oracle.ref_module("superagg")
    cores = os.cpu_count() or 1
    if ref is not None:
        nthreads = cores
        chunk = 1 << 20  # vaex's chunk size cap (vaex/settings.py:83-87)
        pool_threads = [cores]   # (a list: picked below — every core, or fewer where that is faster)

        def one_pass(vals=None):
            nthreads = pool_threads[0]
            vv = vs if vals is None else vals
            bx = ref.BinnerScalar_float64(nthreads, "x", -4.0, 4.0, shape)
            by = ref.BinnerScalar_float64(nthreads, "y", -4.0, 4.0, shape)
            grid = ref.Grid([bx, by])
            aggs = [ref.AggCount_int64(grid, nthreads, nthreads), ref.AggSum_float64(grid, nthreads, nthreads), ref.AggCount_float64(grid, nthreads, nthreads)]
            import queue
            slots = queue.Queue()
            for t in range(nthreads):
                slots.put(t)
            # Touch every thread slot once from THIS thread before the pool starts: the reference marks a thread's grid as used
            # in a std::vector<bool> (src/agg_base.hpp:17,:41-44) — bits of one word written by different threads without
            # synchronisation.  With 256 threads starting together a lost bit makes get_result skip that thread's grid (seen as
            # ~1000 cells short in 2 of 10 runs); after this loop the bits are only read.
            for t in range(nthreads):
                bx.set_data(t, xs[:1]); by.set_data(t, ys[:1])
                bx.clear_data_mask(t); by.clear_data_mask(t)
                aggs[1].set_data(t, vv[:1], 0); aggs[2].set_data(t, vv[:1], 0)
                for a in aggs:
                    a.clear_data_mask(t)
                grid.bin(t, aggs, 0)

            def work(i1):
                t = slots.get()
                try:
                    i2 = min(rows, i1 + chunk)
                    bx.set_data(t, xs[i1:i2]); by.set_data(t, ys[i1:i2])
                    bx.clear_data_mask(t); by.clear_data_mask(t)
                    aggs[1].set_data(t, vv[i1:i2], 0); aggs[2].set_data(t, vv[i1:i2], 0)
                    for a in aggs:
                        a.clear_data_mask(t)
                    grid.bin(t, aggs, i2 - i1)
                finally:
                    slots.put(t)
            t0 = time.perf_counter()  # (object construction and the slot touching above are not part of the pass)
            with ThreadPoolExecutor(nthreads) as pool:
                list(pool.map(work, range(0, rows, chunk)))
            res = [a.get_result() for a in aggs]
            return res, time.perf_counter() - t0
        kind = "reference"
    else:
        nthreads = 1
        case = dict(n=rows, binners=[dict(kind="scalar", data=xs, vmin=-4, vmax=4, bins=shape), dict(kind="scalar", data=ys, vmin=-4, vmax=4, bins=shape)],
                    aggs=[dict(kind="count"), dict(kind="sum", data=vs), dict(kind="count", data=vs)])

        def one_pass():
            t0 = time.perf_counter()
            res = oracle.run_case(case)
            return res, time.perf_counter() - t0
        kind = "port"
    tried = {}
    if ref is not None and cores > 32:
        # the fairest CPU number: a pool of every host core, or of 64 / 32 of them where hundreds of threads over 100 1-Mi-row chunks
        # get in each other's way (vaex's own executor: 3.0e9 rows/s at 32 threads against 3.8e7 at 256 on the round-5 box)
        for t in (cores, 64, 32):
            pool_threads[0] = t
            one_pass()
            tried[t] = min(one_pass()[1] for _ in range(2))
        pool_threads[0] = min(tried, key=tried.get)
        nthreads = pool_threads[0]
    best = float("inf")
    res = None
    t_start = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t_start < 10 and reps < 20):
        res, dt = one_pass()
        best = min(best, dt)
        reps += 1
    sabs = None
    if ref is not None:
        # sum|v| per cell (the scale of the fp64 tolerance, `1e-12 x sum|v|` per cell): one more pass of the reference's AggSum over |v|
        sabs = np.asarray(one_pass(np.abs(vs))[0][1])
    single = None
    if ref is not None:
        # the raw single-thread Grid.bin rate of the reference on 2e7 of the rows (SURVEY §8d asks for it beside the pool's)
        m = min(rows, 20_000_000)
        bx = ref.BinnerScalar_float64(1, "x", -4.0, 4.0, shape); by = ref.BinnerScalar_float64(1, "y", -4.0, 4.0, shape)
        g1 = ref.Grid([bx, by])
        a1 = [ref.AggCount_int64(g1, 1, 1), ref.AggSum_float64(g1, 1, 1), ref.AggCount_float64(g1, 1, 1)]
        bx.set_data(0, xs[:m]); by.set_data(0, ys[:m]); a1[1].set_data(0, vs[:m], 0); a1[2].set_data(0, vs[:m], 0)
        t1 = time.perf_counter()
        g1.bin(0, a1, m)
        single = m / (time.perf_counter() - t1)
    through_vaex = None
    if ref is not None:
        # vaex's executor at VAEX_NUM_THREADS = every host core (north_star's wording) and, where that is more than 32, at 32 as well —
        # its thread pool and per-chunk Python do not scale to hundreds of threads; both are reported, `value` is the better one
        runs = [vaex_cpu_pass(xs, ys, vs, shape, t) for t in sorted({cores, min(cores, 32)}, reverse=True)]
        good = [r for r in runs if r and "value" in r]
        through_vaex = dict(max(good, key=lambda r: r["value"]), runs=[{k: r.get(k) for k in ("value", "threads", "error") if k in r} for r in runs if r]) if good else (runs[0] if runs else None)
    return dict(value=rows / best, unit="rows/s", cores=nthreads, host_cores=cores, pool_sizes_tried={str(t): rows / dt for t, dt in tried.items()} or None, kind=kind, single_thread_value=single, through_vaex=through_vaex,
                sample=f"{rows:.3g} of the GPU's own rows (x,y,v float64), same 2-D {shape}x{shape} count+sum+count pass, best of {reps} passes, 1Mi-row chunks over a {nthreads}-thread pool"), res, rows, sabs


def alloc_probe(torch, when, gb=20):
    """what a 20 GB allocation costs in THIS process right now (a fresh block from the runtime through torch's allocator, handed straight back).  The dense groupby's
    process-first call — the first that asks for the record streams — was 14-16 ms on some runs and 0.5-3.7 s on others, all of it inside ONE hipMalloc (the line's
    `first_call_in_process.pool`); runs with --no-cpu never showed it: the probe is taken behind the configs AND behind the CPU baseline (hundreds of host threads,
    gigabytes of host arrays made and dropped) to say which state of the process makes the runtime's allocator slow."""
    import time as _t
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    tq = _t.perf_counter()
    blk = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t_alloc = (_t.perf_counter() - tq) * 1e3
    del blk
    tq = _t.perf_counter()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    return {"gb": gb, "ms_alloc": round(t_alloc, 3), "ms_free": round((_t.perf_counter() - tq) * 1e3, 3), "when": when}


def other_configs(sa, torch, rows, sample_rows):
    """BASELINE configs[2] (3-D 128^3 histogram with a boolean selection) and configs[3] (groupby on a 1e6-cardinality int64
    key, sum / mean / std — dense keys and scattered keys) at `rows` rows through vaex_amd.binned.Frame, each with its wall
    rate, the HIP-event time of its kernels on the library's stream, a roofline object (SURVEY §8d bytes per row) and a
    same-run parity check of a `sample_rows` slice against the reference's own C++ (oracle/_ref)."""
    from oracle import oracle
    from vaex_amd.binned import Frame, agg
    ref = oracle.ref_module("superagg")
    out = []
    g = torch.Generator(device="cuda").manual_seed(7)

    stream_ms = [0.0]

    first_call = [0.0, 0.0]
    first_detail = [None]
    process_first = [None]
    process_first_detail = [None]

    def timed(fn, reps=3, prime=None, info=None, again=None):
        # `prime`: the same call over COPIES of the columns (other column objects) first, so that what a PROCESS pays once at this size —
        # code objects loaded on first launch, gigabytes of queue scratch and the pinned result buffers allocated (0.5-0.7 s for a
        # 1e9-row groupby: `ms_first_call_in_process`) — is not booked on the columns: `ms_first_call` is what a later call over FRESH
        # columns pays (their key range / NaN scan, their hot-box sample)
        process_first[0] = None
        process_first_detail[0] = None
        # the FIRST call over fresh columns is timed too: it pays what the later ones find remembered per column
        # object — the groupby's exact key-range pass (vxh_minmax_int, 8 B/row) and NaN scan of the value column, the hot-box sample —
        # and goes on the line as `ms_first_call` / `kernel_ms_first_call`; `ms` / `kernel_ms` are the best of the warm calls after it
        def first(label, fn=fn):
            # the first call with a host clock around every library call it makes and the block pool's hipMalloc / hipFree counters around
            # it: a first call far above the warm ones (the driver's round-5 line: 464 ms against 11) then says on the line itself where the
            # time went — a library call's host time, the runtime's allocator, or neither (the device queue itself stalled)
            names = [nm for nm in ("groupby_run", "scan_key_value", "minmax_int", "minmax", "finish") if hasattr(sa, nm)]
            saved, log = {nm: getattr(sa, nm) for nm in names}, []

            def wrap(nm, f):
                def g_(*a, **kw):
                    tw = time.perf_counter()
                    try:
                        return f(*a, **kw)
                    finally:
                        log.append([nm, round((time.perf_counter() - tw) * 1e3, 3)])
                return g_
            pool0 = {k_: sa.config_get(k_) for k_ in ("pool_mallocs", "pool_malloc_bytes", "pool_malloc_us", "pool_frees", "pool_free_us")}
            free0 = torch.cuda.mem_get_info()[0]
            # ... and around the Frame's own steps between those calls (the heavy-key sample is torch.unique over 2^17 keys: torch's allocator and
            # kernels, not this library's), with the number of blocks torch's caching allocator had to ask the runtime for meanwhile
            fnames = [nm for nm in ("_heavy_keys", "_prescan", "_key_range", "_may_hold_nan") if hasattr(Frame, nm)]
            fsaved = {nm: getattr(Frame, nm) for nm in fnames}
            tstats0 = torch.cuda.memory_stats()
            for nm in names:
                setattr(sa, nm, wrap(nm, saved[nm]))
            for nm in fnames:
                setattr(Frame, nm, wrap("Frame." + nm, fsaved[nm]))
            try:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sa.timer_start(0)
                r_ = fn()
                sa.timer_stop(0)
                k_first = sa.timer_kernels_ms(0)
                ms_first = (time.perf_counter() - t0) * 1e3
            finally:
                for nm in names:
                    setattr(sa, nm, saved[nm])
                for nm in fnames:
                    setattr(Frame, nm, fsaved[nm])
            tstats1 = torch.cuda.memory_stats()
            detail = {"label": label, "ms": round(ms_first, 3), "kernel_ms": round(k_first, 3), "library_calls_ms": log,
                      "pool": {k_: sa.config_get(k_) - v_ for k_, v_ in pool0.items()}, "pool_cached_gb": round(sa.config_get("pool_cached_bytes") / 2**30, 2),
                      "hbm_free_gb_before": round(free0 / 2**30, 1),
                      "torch_allocator": {k_: int(tstats1.get(k_, 0)) - int(tstats0.get(k_, 0)) for k_ in ("num_device_alloc", "num_device_free", "num_alloc_retries")}}
            if info is not None:   # (the groupby's own account of THIS call: retries, buckets, kernel times)
                detail["groupby_info"] = {k_: (round(v_, 3) if isinstance(v_, float) else v_) for k_, v_ in (info() or {}).items()}
            del r_
            return ms_first, k_first, detail
        if prime is not None:
            run_primed = prime()   # (the copies are made HERE, outside the clock: torch's allocator asking the runtime for 8 GB blocks was 0.3-0.6 s of round 5's `ms_first_call_in_process`)
            # the same account as the first call's: library calls on the host clock, the block pool's hipMalloc / hipFree counters, torch's allocator — a slow process-first
            # call (the dense groupby: the first call that needs the 24 GB of record streams) then says on the line where its time went
            process_first[0], _, process_first_detail[0] = first("first call of this kind in the process, over copies of the columns", run_primed)
            del run_primed
            torch.cuda.empty_cache()
        first_call[0], first_call[1], first_detail[0] = first("first call over fresh columns")
        best, best_k = float("inf"), float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sa.timer_start(0)
            res = fn()
            s_ms = sa.timer_stop(0)          # start .. everything on the library's stream, result columns crossing PCIe included
            k_ms = sa.timer_kernels_ms(0)    # start .. end of the call's last kernel
            dt = time.perf_counter() - t0
            if k_ms < best_k:
                best_k, stream_ms[0] = k_ms, s_ms
            best = min(best, dt)
        if again is not None and first_call[0] > 3 * best * 1e3:
            # an outlier (round 5's driver line: 464 ms against 11; one builder box of five in round 6): the same first call once more over FRESH
            # columns, so that the line itself says whether it is a property of the call or of that moment — both accounts stay on the line
            first_detail[0]["again_over_fresh_columns"] = again(first)
        return res, best, best_k

    def line(config, what, bytes_per_row, wall, k_ms, kernel, parity):
        gbs = bytes_per_row * rows / (k_ms * 1e-3) / 1e9
        return {"config": config, "what": what, "rows": rows, "rows_per_s": rows / wall, "ms": wall * 1e3, "kernel_ms": k_ms, "stream_ms": stream_ms[0], "kernel": kernel,
                "ms_first_call": first_call[0], "kernel_ms_first_call": first_call[1], "ms_first_call_in_process": process_first[0], "first_call": first_detail[0], "first_call_in_process": process_first_detail[0],
                "roofline": {"bound": "hbm", "bytes_per_row": bytes_per_row, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
                "parity_on_sample": parity}

    m = int(min(sample_rows, rows))
    # ---- configs[2]: 3-D 128^3 count with the selection v > 3 handed over as a byte mask (what vaex materialises) ----
    x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    z = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    sel = (v > 3).to(torch.uint8)
    torch.cuda.synchronize()
    df = Frame(dict(x=x, y=y, z=z, sel=sel))
    lim3 = [[-4, 4]] * 3
    # ---- north_star's target sentence: 2-D count(*) on a 256x256 grid (16 B/row; src/agg_count.cpp:43-67) ----
    c2d, wall, k_ms = timed(lambda: df.count(binby=["x", "y"], limits=lim3[:2], shape=256, edges=True),
                            prime=lambda: (lambda f: lambda: f.count(binby=["x", "y"], limits=lim3[:2], shape=256, edges=True))(Frame(dict(x=x.clone(), y=y.clone()))))
    kernel = sa.last_kernel(0)
    parity = None
    if ref is not None:
        head = Frame(dict(x=x[:m], y=y[:m])).count(binby=["x", "y"], limits=lim3[:2], shape=256, edges=True)
        bs = [ref.BinnerScalar_float64(1, nm, -4.0, 4.0, 256) for nm in "xy"]
        grid = ref.Grid(bs)
        c = ref.AggCount_int64(grid, 1, 1)
        cols = [t[:m].cpu().numpy() for t in (x, y)]
        for b, col in zip(bs, cols):
            b.set_data(0, col); b.clear_data_mask(0)
        c.clear_data_mask(0)
        grid.bin(0, [c], m)
        want = np.asarray(c.get_result())
        parity = {"ok": bool(np.array_equal(np.asarray(head), want)) and int(np.asarray(c2d).sum()) == rows, "sample_rows": m,
                  "cells_differ": int((np.asarray(head) != want).sum()), "rows_counted": [int(np.asarray(c2d).sum()), rows]}
    out.append(line("count2d", "2-D count(*) of float64 x,y on a 256x256 grid (north_star's target sentence)", 16, wall, k_ms, kernel, parity))
    del c2d
    c3, wall, k_ms = timed(lambda: df.count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="sel", edges=True),
                           prime=lambda: (lambda f: lambda: f.count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="sel", edges=True))(Frame(dict(x=x.clone(), y=y.clone(), z=z.clone(), sel=sel.clone()))))
    kernel = sa.last_kernel(0)
    parity = None
    if ref is not None:
        head = Frame(dict(x=x[:m], y=y[:m], z=z[:m], sel=sel[:m])).count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="sel", edges=True)
        cols = [t[:m].cpu().numpy() for t in (x, y, z)]
        keep = sel[:m].cpu().numpy()
        bs = [ref.BinnerScalar_float64(1, nm, -4.0, 4.0, 128) for nm in "xyz"]
        grid = ref.Grid(bs)
        c = ref.AggCount_int64(grid, 1, 1)
        for b, col in zip(bs, cols):
            b.set_data(0, col); b.clear_data_mask(0)
        c.set_data_mask(0, keep)
        grid.bin(0, [c], m)
        want = np.asarray(c.get_result())
        parity = {"ok": bool(np.array_equal(np.asarray(head), want)), "sample_rows": m, "cells_differ": int((np.asarray(head) != want).sum()),
                  "rows_counted": [int(np.asarray(c3).sum()), int(sel.sum().item())]}
        parity["ok"] = parity["ok"] and parity["rows_counted"][0] == parity["rows_counted"][1]
    out.append(line("configs[2]", "3-D 128^3 count(*) of float64 x,y,z with a boolean selection mask", 25, wall, k_ms, kernel, parity))
    # ---- configs[2]': the same histogram with the selection as an EXPRESSION over a fourth column — evaluated inside the binning
    # kernel (no mask bytes: x, y, z, v = 32 B/row; through a separate predicate pass it was 8 + 1 + 24 + 1 = 34) ----
    dfe = Frame(dict(x=x, y=y, z=z, v=v))
    c3e, wall, k_ms = timed(lambda: dfe.count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="v > 3", edges=True),
                            prime=lambda: (lambda f: lambda: f.count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="v > 3", edges=True))(Frame(dict(x=x.clone(), y=y.clone(), z=z.clone(), v=v.clone()))))
    kernel = sa.last_kernel(0)
    parity = None
    if ref is not None:
        heade = Frame(dict(x=x[:m], y=y[:m], z=z[:m], v=v[:m])).count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="v > 3", edges=True)
        parity = {"ok": bool(np.array_equal(np.asarray(heade), want)) and bool(np.array_equal(np.asarray(c3e), np.asarray(c3))), "sample_rows": m,
                  "cells_differ": int((np.asarray(heade) != want).sum()), "equals_the_mask_form_on_all_rows": bool(np.array_equal(np.asarray(c3e), np.asarray(c3))),
                  "selection_fused_in_kernel": bool(sa.config_get("pred_fused") > 0)}
    out.append(line("configs[2]'", "3-D 128^3 count(*) of float64 x,y,z, selection = the expression \"v > 3\" over a fourth float64 column (evaluated in the binning kernel)", 32, wall, k_ms, kernel, parity))
    # ---- configs[2]'': a TWO-term selection over two further float64 columns, "(v > 3) & (w < 1)" — both read by the binning kernel
    # (round 5: PredDesc::col2; x, y, z, v, w = 40 B/row; through the predicate pass it was 16 + 1 + 24 + 1 = 42) ----
    w = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    sel2 = ((v > 3) & (w < 1)).to(torch.uint8)
    dfe2 = Frame(dict(x=x, y=y, z=z, v=v, w=w))
    f_before = sa.config_get("pred_fused")
    c3t, wall, k_ms = timed(lambda: dfe2.count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="(v > 3) & (w < 1)", edges=True))
    kernel = sa.last_kernel(0)
    parity = None
    if ref is not None:
        headt = Frame(dict(x=x[:m], y=y[:m], z=z[:m], v=v[:m], w=w[:m])).count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="(v > 3) & (w < 1)", edges=True)
        c2 = ref.AggCount_int64(grid, 1, 1)
        keep2 = sel2[:m].cpu().numpy()   # (kept alive: the reference borrows the pointer)
        c2.set_data_mask(0, keep2)
        grid.bin(0, [c2], m)
        want2 = np.asarray(c2.get_result())
        via_mask = Frame(dict(x=x, y=y, z=z, sel2=sel2)).count(binby=["x", "y", "z"], limits=lim3, shape=128, selection="sel2", edges=True)
        parity = {"ok": bool(np.array_equal(np.asarray(headt), want2)) and bool(np.array_equal(np.asarray(c3t), np.asarray(via_mask))), "sample_rows": m,
                  "cells_differ": int((np.asarray(headt) != want2).sum()), "equals_the_mask_form_on_all_rows": bool(np.array_equal(np.asarray(c3t), np.asarray(via_mask))),
                  "rows_counted": [int(np.asarray(c3t).sum()), int(sel2.sum().item())], "selection_fused_in_kernel": bool(sa.config_get("pred_fused") > f_before)}
        del via_mask
    out.append(line("configs[2]''", "3-D 128^3 count(*) of float64 x,y,z, selection = \"(v > 3) & (w < 1)\" over two further float64 columns (both evaluated in the binning kernel)", 40, wall, k_ms, kernel, parity))
    del df, dfe, dfe2, x, y, z, w, sel, sel2, c3, c3e, c3t
    # ---- configs[3]: groupby on 1e6 int64 keys, agg sum / mean / std of v ----
    k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
    spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
    for flavour in ("dense", "scattered"):
        keys = k if flavour == "dense" else (k * 2654435761) % (1 << 40)
        torch.cuda.synchronize()
        df = Frame(dict(k=keys, v=v))
        def once_more(first_fn, keys=keys):
            nonlocal_df[0] = Frame(dict(k=keys.clone(), v=v.clone()))
            return first_fn("the same first call again, over fresh clones")[2]
        nonlocal_df = [None]
        res, wall, k_ms = timed(lambda: (nonlocal_df[0] or df).groupby("k", spec), prime=lambda: (lambda f: lambda: f.groupby("k", spec))(Frame(dict(k=keys.clone(), v=v.clone()))),
                                info=lambda: getattr(nonlocal_df[0] or df, "last_groupby_info", None), again=once_more)
        nonlocal_df[0] = None
        info = getattr(df, "last_groupby_info", None) or {}
        # (round 6: the dense range takes the fused pass too — with a direct LDS table; `info` is that pass's own account, absent for the slab-partitioned pair)
        kernel = ("gb_scatter+gb_reduce" + ("_direct" if info.get("direct_table") else "")) if "ms_scatter" in info else sa.last_kernel(0)
        parity = None
        if ref is not None:
            head = Frame(dict(k=keys[:m], v=v[:m])).groupby("k", spec)
            ks, vs = keys[:m].cpu().numpy(), v[:m].cpu().numpy()
            uniq, codes = np.unique(ks, return_inverse=True)
            # the reference's pass 2: BinnerOrdinal over the keys' ordinals + AggSum / AggCount / AggSumMoment (vaex/cpu.py:678-786)
            b = ref.BinnerOrdinal_int64(1, "k", len(uniq), 0, False, False)
            grid = ref.Grid([b])
            aggs = [ref.AggSum_float64(grid, 1, 1), ref.AggCount_float64(grid, 1, 1), ref.AggSumMoment_float64(grid, 1, 1, 2)]
            codes = np.ascontiguousarray(codes.astype(np.int64))
            b.set_data(0, codes); b.clear_data_mask(0)
            for a in aggs:
                a.set_data(0, vs, 0); a.clear_data_mask(0)
            grid.bin(0, aggs, m)
            s1, cnt, s2 = (np.asarray(a.get_result())[:len(uniq)] for a in aggs)
            sabs = np.bincount(codes[vs == vs], weights=np.abs(vs[vs == vs]), minlength=len(uniq))
            with np.errstate(divide="ignore", invalid="ignore"):
                var = s2 / cnt - (s1 / cnt) ** 2
            okv = cnt > 0
            detail = {"keys_equal": bool(np.array_equal(head["k"], uniq)), "count_groups_differ": int((head["c"] != cnt).sum()) if len(head["c"]) == len(cnt) else -1,
                      "sum_groups_over_tol": int((np.abs(head["s"] - s1) > 1e-12 * sabs).sum()) if len(head["s"]) == len(s1) else -1,
                      "var_groups_over_tol": int((np.abs(head["sd"][okv] ** 2 - var[okv]) > 4e-12 * (s2[okv] / cnt[okv]) + 1e-300).sum()) if len(head["sd"]) == len(cnt) else -1,
                      "groups": [int(len(res["k"])), 1_000_000], "rows_counted": [int(res["c"].sum()), rows], "sample_rows": m}
            detail["ok"] = detail["keys_equal"] and not (detail["count_groups_differ"] or detail["sum_groups_over_tol"] or detail["var_groups_over_tol"]) and detail["rows_counted"][0] == rows
            parity = detail
        ln = line("configs[3]" if flavour == "dense" else "configs[3]'", f"groupby on 1e6 {flavour} int64 keys: count / sum / mean / std of float64 v", 16, wall, k_ms, kernel, parity)
        if info:
            ln["groupby_kernels_ms"] = {kk: info[kk] for kk in ("ms_scatter", "ms_reduce", "ms_sort") if kk in info}
            ln["groupby_pass"] = {kk: info[kk] for kk in ("buckets", "retries", "compact_records", "direct_table", "heavy_keys_in_pass", "dense_range_through_fused_pass") if kk in info}
            # what a two-pass partition can reach at all: every row crosses the fabric three times — read (16 B), written as a
            # record, read back as a record — and mixed read / write traffic moves at the chip's copy rate (6.29 TB/s measured, MI355X_MICROARCH.md),
            # not at the 8 TB/s read peak `frac` is quoted against: floor_frac = 16 / structural bytes x 6.29 / 8
            rec = 10 if info.get("direct_table") else (12 if info.get("compact_records") else 16)
            ln["roofline"]["floor"] = {"structural_bytes_per_row": 16 + 2 * rec, "record_bytes": rec, "mixed_traffic_rate_GBs": 6290.0,
                                        "floor_frac": 16.0 / (16 + 2 * rec) * 6290.0 / HBM_PEAK_GBS,
                                        "note": "two-pass partition: 16 B read + record written + record read back per row, at the measured copy rate"}
        if flavour == "dense":
            ln["large_alloc_probe"] = alloc_probe(torch, "behind this config's calls")
        out.append(ln)
        del df, res
    return out


def measured_traffic(rows, shape, timeout=90):
    """HBM bytes per launch of the bench pass from the PMC counters of THIS command on THIS box (the line used to carry a constant
    read from profiles/): two rocprofv3 passes over a short run of this script — `--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`, each with --kernel-trace
    only, as MI355X_MICROARCH.md's HBM section prescribes — FETCH_SIZE counted twice (gfx950 books a 128-byte request as 64), both in KiB.  Per kernel of the
    pass the average per dispatch.  -> (bytes per launch, per-kernel breakdown) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="vxh_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1",
                   "--no-cpu", "--no-extra", "--no-configs", "--rows", str(int(rows)), "--shape", str(int(shape))]
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {ctr}: rc {p.returncode}, {len(files)} counter files"
            acc = {}
            for f in files:
                with open(f, newline="") as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") != ctr:
                            continue
                        name = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
                        a = acc.setdefault(name, [0.0, set()])
                        a[0] += float(r["Counter_Value"])
                        a[1].add(r["Dispatch_Id"])
            per[ctr] = {k: (v[0] / len(v[1]), len(v[1])) for k, v in acc.items()}
        except Exception as e:   # noqa: BLE001  (a profiler that is not there, times out or writes another format: the line says so and keeps the offline figure)
            return None, f"rocprofv3 --pmc {ctr}: {type(e).__name__}: {str(e)[:120]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    total, parts = 0.0, {}
    for prefix in ("part_scatter_wv", "part_reduce_", "part_merge", "part_hot_merge"):   # the kernels of one pass (one dispatch each per launch)
        cands = [(n, k, avg) for k, (avg, n) in per["FETCH_SIZE"].items() if k.startswith(prefix)]
        if not cands:
            continue
        n, k, f = max(cands)   # (the instantiation with the most dispatches: a first call's timed trial launches the other form once)
        w = per["WRITE_SIZE"].get(k, (0.0, 0))[0]
        total += (2.0 * f + w) * 1024.0
        parts[k[:48]] = {"read_bytes": 2.0 * f * 1024.0, "written_bytes": w * 1024.0, "dispatches": n}
    if not parts:
        return None, "no kernel of the pass among the profiled dispatches"
    return total, parts


def _spawned(local_rank, args, port):
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    run(args)


ROWS_CONFIG1 = 1_000_000_000      # BASELINE configs[1]: 1e9 rows on one GPU
ROWS_CONFIG4_SHARD = 1_250_000_000  # BASELINE configs[4]: 1e10 rows over 8 GPUs


def rows_of_rank(args, rank, world):
    """(rows of this rank, scaling, workload name).  Weak scaling by default: every rank bins the same number of rows — configs[1]'s 1e9 at N = 1 (the
    BENCH line), configs[4]'s 1.25e9-row shard at N > 1 (N = 8: the config itself).  --total-rows: one table split over the ranks (strong)."""
    if args.total_rows is not None:
        from vaex_amd.dist import shard_rows
        total = int(args.total_rows)
        i1, i2 = shard_rows(total, rank, world)
        return i2 - i1, "strong", f"{total:.4g}-row float64 x,y,v table row-sharded over {world} GPU(s) (--total-rows; strong scaling)"
    if args.rows is not None:
        rows = int(args.rows)
        return rows, "weak", f"{rows:.4g}-row float64 x,y,v per GPU (--rows)"
    if world == 1:
        return ROWS_CONFIG1, "weak", "1e9-row float64 x,y,v (BASELINE configs[1])"
    what = "BASELINE configs[4]: 1e10 rows row-sharded over 8 GPUs" if world == 8 else f"the 1.25e9-row shard of BASELINE configs[4] on {world} GPUs ({world * 1.25:.4g}e9 rows)"
    return ROWS_CONFIG4_SHARD, "weak", f"1.25e9-row float64 x,y,v per GPU ({what})"


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched as plain `python bench.py --gpus N`: be the launcher (one rank per GPU of this node)
        import socket
        import torch.multiprocessing as mp
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        mp.spawn(_spawned, args=(args, port), nprocs=args.gpus, join=True)
        return
    run(args)


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # stdout carries ONE line, the JSON: librccl prints a version banner on stdout from C stdio when its first communicator comes up
    # (flushed whenever — seen BEHIND the JSON line), so file descriptor 1 is pointed at stderr for the run and the line goes to a
    # private duplicate of the real stdout at the end
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import vaex_amd
    from vaex_amd import dist as vdist
    sa = vaex_amd.superagg
    if not torch.cuda.is_available() or sa.device_count() == 0:
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    # one GPU per rank; under gloo the ranks may outnumber the GPUs (the dry run of the N > 1 path on a one-GPU box) and share them
    device = local_rank if args.backend == "nccl" else local_rank % sa.device_count()
    torch.cuda.set_device(device)
    sa.set_device(device)
    sa.warmup()   # (what vaex_amd.install() does once: the kernels' code objects and thread slot 0 now, not inside the first call — `ms_first_call_in_process` is measured behind it)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    rows, scaling, workload = rows_of_rank(args, rank, world)
    shape = args.shape
    reduce_dev = "cuda" if args.backend == "nccl" else "cpu"   # where the small timing / row-count reductions live

    gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
    x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen)
    y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen)
    v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen) * 2 + 3
    torch.cuda.synchronize()

    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, shape)
    by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, shape)
    grid = sa.Grid([bx, by])
    count = sa.AggCount_int64(grid, 1, 1)      # count(*)
    vsum = sa.AggSum_float64(grid, 1, 1)       # sum(v)
    vcount = sa.AggCount_float64(grid, 1, 1)   # count(v): non-NaN v
    aggs = [count, vsum, vcount]
    bx.set_data(0, x); by.set_data(0, y)
    bx.clear_data_mask(0); by.clear_data_mask(0)
    vsum.set_data(0, v, 0); vcount.set_data(0, v, 0)
    for a in aggs:
        a.clear_data_mask(0)

    kernel_ms = []
    allreduce_ms = []

    def step():
        for a in aggs:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, aggs, rows)                # the hot path: one fused kernel pass
        kernel_ms.append(sa.timer_stop(0))
        if world > 1:
            t_ar = time.perf_counter()
            vdist.allreduce_aggs(aggs)         # RCCL all-reduce of the three grids
            allreduce_ms.append((time.perf_counter() - t_ar) * 1e3)   # (host clock: the call returns when the grids are the global ones)
        c, s, cv = (a.get_result() for a in aggs)
        with np.errstate(divide="ignore", invalid="ignore"):
            mean = s / cv                      # vaex/agg.py:403-416
        return c[2:-1, 2:-1], mean[2:-1, 2:-1]  # edges=False slice (vaex/agg.py:323-335)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    settle_s = float(os.environ.get("VAEX_AMD_BENCH_SETTLE_S", "0"))   # (experiments: the pass repeated for this long before the warm-up)
    settle_trace = []
    if settle_s > 0:
        t_s = time.perf_counter()
        while time.perf_counter() - t_s < settle_s:
            step()
            settle_trace.append((round(time.perf_counter() - t_s, 3), round(float(kernel_ms[-1]), 3)))
    for _ in range(args.warmup):
        step()
    kernel_ms.clear()
    allreduce_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        c, mean = step()
    barrier()
    elapsed = time.perf_counter() - t0
    total_rows = rows
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tr = torch.tensor([rows], dtype=torch.int64, device=reduce_dev)
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        total_rows = int(tr.item())
    assert int(count.get_result().sum()) == total_rows, "count conservation violated"
    main_kernel = sa.last_kernel(0)

    # ---- N > 1: north_star's target sentence (2-D count(*) on 256x256, 16 B/row) on the same shards, the same clock and the same reduce ----
    count2d = None
    if world > 1 and not args.no_configs:
        c2 = sa.AggCount_int64(grid, 1, 1)
        c2.clear_data_mask(0)
        k2 = []

        def step2():
            c2.reset()
            sa.timer_start(0)
            grid.bin(0, [c2], rows)
            k2.append(sa.timer_stop(0))
            vdist.allreduce_aggs([c2])
            return c2.get_result()
        for _ in range(max(1, args.warmup)):
            step2()
        k2.clear()
        barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            r2 = step2()
        barrier()
        el2 = time.perf_counter() - t2
        t = torch.tensor([el2], dtype=torch.float64, device=reduce_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el2 = float(t.item())
        assert int(r2.sum()) == total_rows, "count conservation violated (count2d)"
        k2_ms = float(np.mean(k2))
        count2d = {"config": "count2d", "what": f"2-D count(*) of float64 x,y on a {shape}x{shape} grid (north_star's target sentence), row-sharded x{world}, one all-reduce of the grid per step",
                   "rows": total_rows, "rows_per_s": total_rows * args.steps / el2, "ms": el2 / args.steps * 1e3, "kernel_ms": k2_ms, "kernel": sa.last_kernel(0),
                   "roofline": {"bound": "hbm", "bytes_per_row": 16, "achieved": 16 * rows / (k2_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": 16 * rows / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_incl_allreduce": 16 * total_rows / (el2 / args.steps) / 1e9 / (world * HBM_PEAK_GBS),
                                "note": "frac: rank 0's kernels against one GPU's peak; frac_incl_allreduce: the whole step on the slowest rank against all GPUs' peak"}}
        del c2

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))
        achieved = BYTES_PER_ROW * rows / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters of the same command (separate rocprofv3 --pmc passes,
        # FETCH_SIZE corrected x2 on gfx950): measured offline, kept under profiles/
        traffic = traffic_source = None
        for tname in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tpath) and main_kernel.startswith("part_scatter") and shape == 256:
                traffic = json.load(open(tpath))["hbm_bytes_per_row"] * rows
                traffic_source = "profiles/" + tname + " (rocprofv3 --pmc passes of this command, FETCH_SIZE x2 on gfx950; not a same-run counter)"
                break
        traffic_parts = None
        if world == 1 and not args.no_extra and traffic is not None and not os.environ.get("VAEX_AMD_BENCH_NO_PMC"):
            # the counters of THIS command on THIS box (two short profiled runs of this script, ~20 s each), when rocprofv3 is there
            t0 = time.perf_counter()
            live, parts = measured_traffic(rows, shape)
            if live is not None:
                traffic, traffic_parts = live, parts
                traffic_source = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (--kernel-trace only) of `bench.py --steps 2 --warmup 1` run by this process on this box, "
                                  f"FETCH_SIZE x2 on gfx950, per-dispatch averages of the pass's kernels ({time.perf_counter() - t0:.0f} s)")
            else:
                traffic_source += f"; a same-run measurement was tried: {parts}"
        out = {
            "metric": "rows/sec, 2-D count+mean on 256x256 grid (count(*), sum(v), count(v) fused), float64 x,y,v",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (device-generated N(0,1) x,y; N(3,2) v; limits [-4,4])",
            "config": {"workload": f"{workload}: count+sum+mean on {shape}x{shape} grid, HBM-resident",
                       "rows_per_gpu": rows, "total_rows": total_rows, "shape": shape, "kernel": main_kernel, "backend": args.backend if world > 1 else None,
                       "parallelism": (f"row-sharded x{world}, " + ("RCCL all-reduce of 3 grids (vxh_allreduce)" if args.backend == "nccl" else "gloo all-reduce of 3 grids through host buffers (dry run: ranks may share a GPU)")) if world > 1 else "one GPU: nothing to reduce"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_of_measured_copy_rate": achieved / 6290.0,  # (6.29 TB/s float4 copy: MI355X_MICROARCH.md)
                         "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": k_ms, "bytes_per_row": BYTES_PER_ROW, "rows_per_launch": rows},
        }
        if traffic is not None:
            out["roofline"]["traffic_bytes_per_row"] = traffic / rows
        if traffic_parts:
            out["roofline"]["traffic_kernels"] = traffic_parts
        if os.environ.get("VAEX_AMD_BENCH_STEPS_DEBUG"):
            out["kernel_ms_per_step"] = [round(float(k), 3) for k in kernel_ms]
            if settle_trace:
                out["settle_trace"] = settle_trace[::max(1, len(settle_trace) // 60)]
        if world > 1:
            # rank 0's kernel time above excludes the reduce; this one is the whole step on the slowest rank against all GPUs' peak
            out["roofline"]["frac_incl_allreduce"] = BYTES_PER_ROW * total_rows / (elapsed / args.steps) / 1e9 / (world * HBM_PEAK_GBS)
            out["roofline"]["allreduce_ms"] = float(np.mean(allreduce_ms))
            out["rccl_ranks"] = dist.get_world_size()
            out["scaling_note"] = ("weak scaling: every rank bins its own rows_per_gpu rows" if scaling == "weak" else "strong scaling: one table split over the ranks") + "; one all-reduce per grid and step"
        if world == 1 and not args.no_extra:
            extra_steps = max(3, min(args.steps, 5))

            def timed(nsteps):
                kernel_ms.clear()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nsteps):
                    step()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                return rows * nsteps / dt, float(np.mean(kernel_ms))
            # (0) the same step 200 more times in a row (~1 s of device time): the timed region above is 0.1 s at the driver's 20 steps —
            # too short for an outside utilisation sampler to see — and a steady second says whether the rate holds
            vs, ks = timed(200)
            out["sustained"] = {"steps": 200, "value": vs, "kernel_ms": ks, "frac": BYTES_PER_ROW * rows / (ks * 1e-3) / 1e9 / HBM_PEAK_GBS}
            # (1) cold call: the hot box is sampled and searched again in every step (a first df.mean on fresh columns)
            sa.config_set("hot_cache", 0)
            step()
            vc, kc = timed(extra_steps)
            sa.config_set("hot_cache", 1)
            out["value_cold"] = vc
            out["roofline"]["frac_cold"] = BYTES_PER_ROW * rows / (kc * 1e-3) / 1e9 / HBM_PEAK_GBS
            # (2) uniform x,y: no cell rectangle is hot — the densest box holds ~15 % of the rows, 85 % go through the partition queues
            gen_u = torch.Generator(device="cuda").manual_seed(4321)
            xu = torch.rand(rows, dtype=torch.float64, device="cuda", generator=gen_u) * 8 - 4
            yu = torch.rand(rows, dtype=torch.float64, device="cuda", generator=gen_u) * 8 - 4
            bx.set_data(0, xu); by.set_data(0, yu)
            step()
            vu, ku = timed(extra_steps)
            out["value_uniform"] = vu
            out["roofline"]["frac_uniform"] = BYTES_PER_ROW * rows / (ku * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["config"]["kernel_uniform"] = sa.last_kernel(0)
            assert int(count.get_result().sum()) == rows
            bx.set_data(0, x); by.set_data(0, y)
            del xu, yu
        if world == 1:
            # The driver's 1-GPU run never reaches the collective of the N > 1 step, and no multi-GPU node has run this repo
            # (no scaling curve has been measured in any round).  So the same C-ABI call — vxh_comm_init + vxh_allreduce, RCCL on
            # the library's stream — runs here on a ONE-rank communicator over the three grids of the last step, outside every
            # timed region: the code path is exercised on the box the bench runs on, its cost at world 1 is on the line, and the
            # grids must come back unchanged.  (The cross-rank merge it stands for: vaex/cpu.py:788-796, src/agg_count.cpp:15-23.)
            try:
                before = [np.array(a.get_result()) for a in aggs]
                comm1 = sa.Comm(1, 0, sa.comm_unique_id())
                comm1.allreduce(aggs)                  # (first call: communicator warm-up)
                torch.cuda.synchronize()
                ar = []
                for _ in range(5):
                    sa.timer_start(0)
                    comm1.allreduce(aggs)
                    ar.append(sa.timer_stop(0))
                same = all(np.array_equal(b, np.array(a.get_result()), equal_nan=True) for b, a in zip(before, aggs))
                out["allreduce_world1_ms"] = float(np.median(ar))
                out["allreduce_world1"] = {"ranks": comm1.size, "grids": len(aggs), "bytes_per_grid": int(before[0].nbytes), "grids_unchanged": bool(same),
                                           "note": "vxh_allreduce (RCCL) on a one-rank communicator, HIP events on the library's stream, outside the timed value; "
                                                   "N > 1 has never run on hardware for this repo: no scaling curve measured"}
                del comm1
            except Exception as e:   # (the bench line must not depend on it)
                out["allreduce_world1"] = {"error": repr(e)}
        if count2d is not None:
            out["configs"] = [count2d]
        if world == 1 and not args.no_configs:
            # (in front of the CPU baseline since late round 6: behind it the runtime's allocator was slow — a 20 GB hipMalloc 2.3-3.7 s instead of 0.4 ms, see
            #  alloc_probe — and the configs' process-first calls were measuring that; the bench's own columns stay where they are: 24 of 288 GB)
            out["configs"] = other_configs(sa, torch, rows, 1e7)
        if not args.no_cpu:
            # (at N > 1 too: rank 0's own shard is the sample — the other ranks wait at the end of the job, outside every timed region)
            cb, cpu_res, cpu_rows, sabs = cpu_baseline(x, y, v, shape, args.cpu_rows if world == 1 else min(args.cpu_rows, 5e7))
            cb["driver"] = "value: Grid.bin of the reference's C++ over a bare thread pool (no vaex executor / expression layer on top: slightly favours the CPU); through_vaex: the same pass through the real vaex package (its executor, VAEX_NUM_THREADS = cores)"
            out["cpu_baseline"] = cb
            # same-run parity on the CPU sample: counts bit-exact, sums to 1e-12 x sum|v| of the cell
            for a in aggs:
                a.reset()
            bx.set_data(0, x[:cpu_rows]); by.set_data(0, y[:cpu_rows]); vsum.set_data(0, v[:cpu_rows], 0); vcount.set_data(0, v[:cpu_rows], 0)
            grid.bin(0, aggs, cpu_rows)
            g = [a.get_result() for a in aggs]
            if sabs is not None:
                bad_sum = np.abs(g[1] - cpu_res[1]) > 1e-12 * sabs  # north_star's bound, per cell
                out["cpu_baseline"]["sum_tolerance"] = "1e-12 x sum|v| per cell (sum|v| from the reference's AggSum over |v|)"
            else:
                vmax = float(torch.nan_to_num(v[:cpu_rows]).abs().max().item())
                bad_sum = np.abs(g[1] - cpu_res[1]) > 1e-12 * vmax * np.maximum(g[2], 1)  # per cell: 1e-12 * (>= sum|v| of the cell)
                out["cpu_baseline"]["sum_tolerance"] = "1e-12 x max|v| x count per cell (C port: no sum|v| pass)"
            detail = {"count_cells_differ": int((g[0] != cpu_res[0]).sum()), "countv_cells_differ": int((g[2] != cpu_res[2]).sum()),
                      "sum_cells_over_tol": int(bad_sum.sum()), "rows_counted": [int(g[0].sum()), int(cpu_res[0].sum())]}
            out["cpu_baseline"]["parity_on_sample"] = not any(detail[k] for k in ("count_cells_differ", "countv_cells_differ", "sum_cells_over_tol"))
            out["cpu_baseline"]["parity_detail"] = detail
            if world == 1:
                out["large_alloc_probe_behind_cpu_baseline"] = alloc_probe(torch, "behind the CPU baseline")
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
