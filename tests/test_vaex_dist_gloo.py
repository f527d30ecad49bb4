"""install(distributed=True) without a GPU: two processes (gloo), each running the SAME program of the real vaex package on its own
row shard, the task parts' reduce() merging across the ranks (vaex_amd/vaex_dist.py) — compared with one process over the whole
table.  Local compute is vaex's own C++ (the product has no CPU kernels): what this covers is where the cross-rank merge sits and
what it merges (aggregator grids through their host buffers, legacy statistic grids through the task's own op.reduce, the device
groupby's Frame getting the communicator, tasks without a cross-rank form refusing).  The -m gpu twin runs two HIP processes."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAEXPY = os.path.join(ROOT, "oracle", "_ref", "vaexpy")
OVERLAY = os.path.join(ROOT, "oracle", "_ref", "overlay")
FAKE = os.path.join(ROOT, "oracle", "fake")
PKG = VAEXPY if os.path.isdir(os.path.join(VAEXPY, "vaex")) else OVERLAY

SCRIPT = r'''
import os, sys, numpy as np
sys.path[:0] = [%(pkg)r, %(fake)r, %(root)r]
rank, world, gpu = int(sys.argv[1]), int(sys.argv[2]), %(gpu)d
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[3], RANK=str(rank), WORLD_SIZE=str(world))
import torch.distributed as dist
import vaex, vaex_amd
from vaex_amd import vaex_dist, vaex_groupby as vg, binned
dist.init_process_group("gloo", rank=rank, world_size=world)   # (two HIP processes on ONE GPU cannot form an RCCL group: the grids travel through their host buffers)
rng = np.random.default_rng(11)
n = 120_000
v = rng.normal(3, 2, n); v[::89] = np.nan
cols = dict(x=rng.normal(0, 1, n), y=rng.normal(0, 1, n), v=v, k=np.sort(rng.integers(-3, 40, n)), i=rng.integers(-100, 100, n).astype("i4"), f4=rng.normal(0, 1, n).astype("f4"))
whole = vaex.from_arrays(**cols)
A = vaex.agg
calls = {
  "count2d": lambda d: d.count(binby=["x", "y"], limits=[[-4, 4], [-4, 4]], shape=32),
  "mean_sel": lambda d: d.mean("v", binby="x", limits=[-3, 3], shape=16, selection="y > 0"),
  "std": lambda d: d.std("v", binby=["x"], limits=[-3, 3], shape=8),
  "minmax": lambda d: d.minmax("v"),
  "limits_minmax": lambda d: d.count(binby="f4", limits="minmax", shape=12),
  "sum_int": lambda d: d.sum("i", binby="y", limits=[-2, 2], shape=5),
  "min_max_binned": lambda d: [d.min("v", binby="x", limits=[-3, 3], shape=6), d.max("f4", binby="x", limits=[-3, 3], shape=6)],
  "count_scalar": lambda d: d.count("v"),
  "pct": lambda d: d.percentile_approx("x", 50),
}
# ... and random ones (round 5): statistic x column x 0-2 binby dimensions x random shapes x selection (none / expression / list), some of them
# delayed in pairs — every rank draws the same calls, in the same order
LIM = dict(x=[-3, 3], y=[-3, 3], v=[-3, 9], f4=[-2.5, 2.5], i=[-100.5, 99.5], k=[-3.5, 39.5])
def draw(seed):
    r = np.random.default_rng(900 + seed)
    stat = str(r.choice(["count", "sum", "mean", "std", "min", "max", "minmax", "count_star"]))
    value = str(r.choice(["x", "y", "v", "i", "f4"]))
    nd = 0 if stat == "minmax" else int(r.choice([0, 1, 1, 2]))
    binby = [str(b) for b in r.choice(list(LIM), size=nd, replace=False)]
    kw = dict(binby=binby, limits=[LIM[b] for b in binby], shape=[int(r.integers(1, 40)) for _ in binby]) if nd else {}
    s = r.random()
    sel = None if s < 0.45 else (str(r.choice(["y > 0", "(v > 2) & (x < 1)", "i != 3", "f4 <= 0.3", "x + y > 0.5"])) if s < 0.85 else [None, "y < 0.25"])
    if sel is not None:
        kw["selection"] = sel
    return stat, value, kw, bool(r.random() < 0.3)
def random_calls(d):
    out, pending = [], []
    def flush():
        if pending:
            d.execute()
            out.extend(p.get() for p in pending)
            del pending[:]
    for seed in range(%(nrandom)d):
        stat, value, kw, delayed = draw(seed)
        fn = (lambda **k: d.count(**k)) if stat == "count_star" else (lambda **k: getattr(d, stat)(value, **k))
        if delayed:
            pending.append(fn(delay=True, **kw))
            if len(pending) == 2:
                flush()
        else:
            flush()
            out.append(fn(**kw))
    flush()
    return out
want_random = random_calls(whole)
want = {name: fn(whole) for name, fn in calls.items()}                       # before anything is installed: plain vaex, the whole table
want_g = whole.groupby("k", agg={"s": A.sum("v"), "c": A.count(), "m": A.mean("v")}, sort=True)
want_g = {c: want_g[c].to_numpy() for c in want_g.get_column_names()}
if gpu:
    assert vaex_amd.superagg.device_count() > 0
    vaex_amd.install(distributed=True)
else:
    from tests.test_golden_api import RefAdapter
    state = {}
    vaex_dist.install(vaex, state)                                          # (vaex's own task parts under the cross-rank reduce)
    ref = RefAdapter(vaex.superagg)
    vg._frame_for = lambda df, columns: binned.Frame(dict(columns), chunk_size=50_000, nthreads=2, superagg=ref, comm=vaex_dist.comm())
    vg.install(vaex, state)
df = vaex_amd.shard(whole)
assert len(df) < len(whole) and vaex_dist.active()
def close(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind in "iub":
        assert np.array_equal(a, b), what
    else:
        assert np.allclose(a, b, rtol=1e-9 if "std" in what else 1e-12, atol=1e-12 * max(1.0, float(np.nanmax(np.abs(b[np.isfinite(b)]))) if np.isfinite(b).any() else 1.0), equal_nan=True), (what, a, b)
for name, fn in calls.items():
    got = fn(df)
    if isinstance(got, list):
        for g, w in zip(got, want[name]):
            close(g, w, name)
    else:
        close(got, want[name], name)
    print("ok", name, flush=True)
got_random = random_calls(df)
assert len(got_random) == len(want_random) == %(nrandom)d
for seed, (g_, w_) in enumerate(zip(got_random, want_random)):
    stat = draw(seed)[0]
    a_, b_ = np.ma.filled(np.ma.asarray(g_).astype("f8"), np.nan), np.ma.filled(np.ma.asarray(w_).astype("f8"), np.nan)
    assert a_.shape == b_.shape, (seed, draw(seed))
    if stat in ("count", "count_star", "min", "max", "minmax"):
        assert np.array_equal(a_, b_, equal_nan=True), (seed, draw(seed))
    else:
        fin = np.abs(b_[np.isfinite(b_)])
        assert np.allclose(a_, b_, rtol=1e-7 if stat == "std" else 1e-11, atol=(1e-6 if stat == "std" else 1e-11) * max(1.0, float(fin.max()) if fin.size else 1.0), equal_nan=True), (seed, draw(seed))
print("ok random", len(got_random), flush=True)
vg.last.clear()
g = df.groupby("k", agg={"s": A.sum("v"), "c": A.count(), "m": A.mean("v")}, sort=True)
assert vg.last.get("path") == "device", vg.last
for c in want_g:
    close(np.ma.getdata(g[c].to_numpy()), np.ma.getdata(want_g[c]), "groupby " + c)
print("ok groupby", flush=True)
for what, fn in (("nunique", lambda: df.i.nunique()), ("unique", lambda: df.unique("k")), ("first", lambda: df.first("v", "i", binby="x", limits=[-3, 3], shape=4))):
    try:
        fn()
        raise SystemExit(what + " answered from one shard")
    except NotImplementedError as e:
        assert "cross-rank" in str(e), e
    print("ok refused", what, flush=True)
assert vaex_dist.stats["aggregations"] >= 8 and vaex_dist.stats["statistics"] >= 2, vaex_dist.stats
dist.barrier()
dist.destroy_process_group()
print("DONE", flush=True)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(gpu, timeout):
    port = _free_port()
    env = dict(os.environ, VAEX_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("VAEX_HOME", "/tmp/vaex_home_dist")
    procs = [subprocess.Popen([sys.executable, "-c", SCRIPT % dict(pkg=PKG, fake=FAKE, root=ROOT, gpu=gpu, nrandom=60), str(r), "2", str(port)], cwd="/tmp", env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0 and "DONE" in o, o[-2000:] + e[-5000:]
        assert o.count("ok refused") == 3 and "ok groupby" in o and "ok random 60" in o, o
    return outs


@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_two_ranks_of_real_vaex_merge_in_reduce():
    _run(0, 600)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_two_hip_ranks_of_real_vaex_merge_in_reduce(gpu_ready):
    _run(1, 900)
