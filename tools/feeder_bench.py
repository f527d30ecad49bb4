#!/usr/bin/env python3
"""Host-streamed binning rates: count(binby=[x, y], shape=256) over HOST numpy columns, chunked the way
vaex's executor does (1 Mi rows per Grid.bin call, one slot per thread).

  plain      knob feeder=0: hipMemcpyAsync from the caller's pageable memory on the compute stream (round 1)
  stream     feeder=1 (default): the same copies on the slot's copy stream into a ring of arenas, kernels event-chained
  ring       feeder=2: CPU copy into the slot's page-locked ring first, then as above (the call returns before the DMA)
  pinned     columns registered with VXH_CACHE_PIN, cache budget 0: DMA straight from the page-locked columns
  cached#1   registered, first pass (fills the device column cache)
  cached#2   second pass over the same columns: chunks served from HBM

usage: feeder_bench.py [rows] [threads,threads,...] [only-mode] [chunk rows, chunk rows, ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vaex_amd  # noqa: E402
from vaex_amd import superagg as sa  # noqa: E402
from vaex_amd.binned import Frame  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
threads = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8]
only = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "all" else None
chunks = [int(float(c)) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1 << 20]
rng = np.random.default_rng(0)
x = rng.standard_normal(n)
y = rng.standard_normal(n)
lim = [[-4, 4], [-4, 4]]
want = None


def run(f, reps=3):
    global want
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        c = f.count(binby=["x", "y"], limits=lim, shape=256)
        best = min(best, time.perf_counter() - t0)
        if want is None:
            want = c
        assert np.array_equal(c, want)
    return best


def line(mode, nt, t):
    print(f"FEEDER {mode:<9} threads={nt:<2} chunk={chunk:<9} rows={n} {t*1e3:9.1f} ms  {n/t/1e9:6.2f} Grows/s  {n*16/t/1e9:6.1f} GB/s of columns", flush=True)


for nt, chunk in [(nt, chunk) for chunk in chunks for nt in threads]:
    f = Frame(x=x, y=y, nthreads=nt, chunk_size=chunk)
    if only in (None, "plain"):
        sa.config_set("feeder", 0)
        line("plain", nt, run(f))
    if only in (None, "ring"):
        sa.config_set("feeder", 2)
        line("ring", nt, run(f))
    sa.config_set("feeder", 1)
    if only in (None, "stream"):
        line("stream", nt, run(f))
    if only in (None, "pinned"):
        sa.config_set("cache_bytes", 0)
        t0 = time.perf_counter()
        vaex_amd.cache_columns({"x": x, "y": y}, pin=True)
        t_reg = time.perf_counter() - t0
        line("pinned", nt, run(f))
        vaex_amd.uncache_columns()
        print(f"FEEDER (hipHostRegister of {2*x.nbytes/1e9:.1f} GB took {t_reg*1e3:.0f} ms)")
    if only in (None, "cached"):
        sa.config_set("cache_bytes", 64 << 30)
        vaex_amd.cache_columns({"x": x, "y": y}, pin=True)
        line("cached#1", nt, run(f, reps=1))
        line("cached#2", nt, run(f))
        print("FEEDER cache", sa.cache_stats())
        vaex_amd.uncache_columns()
