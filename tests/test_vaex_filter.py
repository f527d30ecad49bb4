"""Filtered frames of the REAL vaex (oracle/_ref/vaexpy) under vaex_amd.install(): binned aggregations take the filter as a keep-mask
over uncompacted chunks (vaex_amd/vaex_filter.py) instead of vaex's per-chunk numpy compaction (vaex/execution.py:515-523) — the
results must be those of vaex's own pre-filtered path, computed in the same process before install().

  * here (no GPU): the HIP classes are switched off, so every task part falls back to vaex's own C++ classes — this pins the host
    logic (Run wrap, spec flag, mask AND-ing, mixed runs left alone) bit for bit;
  * `-m gpu`: the product — device predicates for filters in the comparison subset (alone and combined with a device selection),
    host masks for the rest."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAEXPY = os.path.join(ROOT, "oracle", "_ref", "vaexpy")
OVERLAY = os.path.join(ROOT, "oracle", "_ref", "overlay")
FAKE = os.path.join(ROOT, "oracle", "fake")
PKG = VAEXPY if os.path.isdir(os.path.join(VAEXPY, "vaex")) else OVERLAY

SCRIPT = r'''
import sys, numpy as np
sys.path[:0] = [%(pkg)r, %(fake)r, %(root)r]
GPU = %(gpu)d
import vaex, vaex_amd
from vaex_amd import vaex_filter, vaex_selection
rng = np.random.default_rng(11)
n = %(n)d
v = rng.normal(3, 2, n); v[::997] = np.nan
m = np.ma.array(rng.normal(0, 1, n), mask=rng.random(n) < 0.05)
df = vaex.from_arrays(x=rng.normal(0, 1, n), y=rng.normal(0, 1, n), v=v, k=rng.integers(0, 9, n), m=m,
                      f4=rng.choice(np.array([0.1, 0.3, 0.30000001, 0.5, 0.7], dtype="f4"), n))
lim2 = [[-4, 4], [-4, 4]]

def frames():
    return {
        "one_term": df[df.x > 0.5],                                   # device predicate
        "chain": df[df.x > -1][df.y < 1.5],                           # df[a][b]: mode "and" -> (a) & (b)
        "or_not": df[(df.x > 1) | ~(df.y >= 0)],
        "f4_boundary": df[df.f4 <= 0.3],                              # float32 column against a double constant
        "int_key": df[df.k != 3],
        "arithmetic": df[(df.x ** 2 + df.y ** 2) < 4],                # round 5: arithmetic over float64 columns is a device predicate too
        "libm": df[np.sin(df.x) + df.y ** 3 < 0.5],                   # outside the subset (libm, other powers): the executor's host mask
        "five_terms": df[(df.x > -2) & (df.x < 2) & (df.y > -2) & (df.y < 2) & (df.v > 0)],   # too many terms: host mask
        "dropnan": df.dropnan(column_names=["v"]),                    # SelectionDropNa: host mask
        "masked_dep": df[df.m > 0],                                   # a filter over a column with missing values: host mask
    }

def calls(d):
    out = {}
    out["count2d"] = d.count(binby=[d.x, d.y], limits=lim2, shape=32)
    out["mean2d"] = d.mean(d.v, binby=[d.x, d.y], limits=lim2, shape=32)
    out["sum_sel"] = d.sum(d.v, binby=[d.x], limits=[-4, 4], shape=64, selection="y < 0.25")       # device selection next to the filter
    out["std"] = d.std(d.v, binby=[d.y], limits=[-4, 4], shape=16)
    out["minmax"] = np.array([d.min(d.v, binby=[d.x], limits=[-4, 4], shape=16), d.max(d.v, binby=[d.x], limits=[-4, 4], shape=16)])
    out["sel_list"] = d.count(binby=[d.x], limits=[-4, 4], shape=16, selection=[None, "y > 0", "(x > 0) & (v < 2)"])
    out["sel_big"] = d.count(binby=[d.x], limits=[-4, 4], shape=16, selection="(x > -3) & (x < 3) & (y > -3) & (y < 3) & (v > -9)")  # host-evaluated selection
    out["masked_value"] = d.sum(d.m, binby=[d.x], limits=[-4, 4], shape=16)
    out["zero_d"] = np.array([float(d.count()), float(d.count(d.v)), float(d.sum(d.v)), float(d.count(selection="y < 0"))])
    out["ordinal"] = d.count(binby=[d.k], limits=[-0.5, 8.5], shape=9)
    out["minmax_limits"] = d.mean(d.v, binby=[d.x], limits="minmax", shape=8)    # a minmax run (left pre-filtered) before the aggregation run
    d.select("v > 3.5")
    out["named_selection"] = d.count(binby=[d.y], limits=[-4, 4], shape=16, selection=True)          # a named selection: resolved to its expression, a device predicate next to the filter
    d.select_nothing()
    out["first"] = d.first(d.v, d.y, binby=[d.k], limits=[-0.5, 8.5], shape=9)   # AggFirst: the run stays pre-filtered
    g = d.groupby("k", agg={"c": "count", "s": vaex.agg.sum("v")}, sort=True)    # vaex's two passes (the distinct-key pass is pre-filtered)
    out["groupby"] = np.array([g.k.values, g.c.values, g.s.values], dtype="f8")
    out["length"] = np.array([len(d)])
    return out

want = {name: calls(d) for name, d in frames().items()}      # plain vaex: pre-filtered chunks, its own C++

if GPU:
    backend = vaex_amd.install()
else:
    backend = vaex_amd.install(hash_sets=False, legacy=False, groupby=False)
    class _NoHip:
        def __getattr__(self, name):
            raise NotImplementedError("test: HIP classes switched off")
    backend.__dict__["_hip"] = _NoHip()
used = []
_task = vaex_amd._installed["task_hip"]
_decode = _task.decode.__func__
def _recording_decode(cls, *a, **k):
    part = _decode(cls, *a, **k)
    used.append((part.backend_used, part._hip_filter_as_mask, sorted(part._hip_filter_on_device)))
    return part
_task.decode = classmethod(_recording_decode)

report = {}
for name, d in frames().items():
    before = dict(vaex_filter.stats); del used[:]
    got = calls(d)
    delta = {k: vaex_filter.stats[k] - before[k] for k in before}
    report[name] = (delta, list(used))
    for call, w in want[name].items():
        g = got[call]
        w, g = np.asarray(w, dtype="f8"), np.asarray(g, dtype="f8")
        assert w.shape == g.shape, (name, call, w.shape, g.shape)
        assert np.array_equal(np.isnan(w), np.isnan(g)), (name, call)
        if call in ("count2d", "sel_list", "sel_big", "ordinal", "named_selection", "minmax", "first", "length"):
            assert np.array_equal(np.nan_to_num(w), np.nan_to_num(g)), (name, call, w, g)
        else:   # float sums: vaex's pool threads pick chunks up in whatever order they come, the device adds in its own
            tol = 1e-9 if call == "std" else 1e-12 * 10.0 * n
            assert np.all(np.abs(np.nan_to_num(w) - np.nan_to_num(g)) <= tol), (name, call, np.nanmax(np.abs(w - g)))
    # every aggregation run of this frame took the keep-mask form, except the runs holding other kinds of tasks
    assert delta["runs_switched"] >= 10, (name, delta)
    assert delta["runs_mixed"] >= 2, (name, delta)       # AggFirst / AggNUnique / the minmax run stay with vaex's compaction
    parts_as_mask = [u for u in used if u[1]]
    assert parts_as_mask, name
    assert all(u[0] == ("hip" if GPU else "cpu") for u in parts_as_mask), (name, used)
    device_filter = name in ("one_term", "chain", "or_not", "f4_boundary", "int_key", "arithmetic")
    if GPU and device_filter:
        assert delta["device_chunks"] > 0, (name, delta)
        assert any(u[2] for u in parts_as_mask), (name, used)
    else:
        assert delta["device_chunks"] == 0 and delta["host_chunks"] > 0, (name, delta)
    print(name, delta, flush=True)
# a frame with functions of its own: the filter may be what protects the function from rows it cannot take — the reference's own test,
# /root/reference/tests/agg_test.py:405-416 and tests/selection_test.py:167-177 — so such frames keep vaex's compaction
def custom_func(x):
    assert 4 not in x; return x**2
dfu = vaex.from_arrays(x=np.arange(10))
dff = dfu[dfu.x != 4]
dff.add_function('custom_function', custom_func)
dff['y'] = dff.func.custom_function(dff.x)
before = dict(vaex_filter.stats)
assert dff.count(dff.y) == 9 and dff.count(dff.y, selection='y > 0') == 8
assert vaex_filter.stats["runs_switched"] == before["runs_switched"], (before, vaex_filter.stats)
print("user-function frame keeps the compaction", flush=True)
# an unfiltered frame is nobody's business here
before = dict(vaex_filter.stats)
calls(df)
assert vaex_filter.stats == before
vaex_amd.uninstall()
import vaex.execution, vaex.tasks
assert vaex.execution.Run.__init__.__qualname__.startswith("Run.") and vaex.tasks.TaskAggregations.encode.__qualname__.startswith("TaskAggregations.")
again = calls(frames()["one_term"])
assert np.allclose(np.nan_to_num(again["mean2d"]), np.nan_to_num(want["one_term"]["mean2d"]), rtol=1e-12, atol=0) and np.array_equal(again["count2d"], want["one_term"]["count2d"])
print("FILTER OK", flush=True)
'''


def _run(gpu, n):
    if not os.path.isdir(os.path.join(PKG, "vaex")):
        pytest.skip("oracle/_ref/vaexpy not built (run __graft_entry__.build() where /root/reference exists)")
    env = dict(os.environ)
    env.setdefault("VAEX_HOME", "/tmp/vaex_home_filter")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(pkg=PKG, fake=FAKE, root=ROOT, gpu=gpu, n=n)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "FILTER OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_filtered_frames_host_logic_on_vaex_cpp():
    _run(0, 200_000)


@pytest.mark.gpu
def test_filtered_frames_on_the_device():
    _run(1, 3_000_000)


CHAIN_FUZZ = r"""
import sys, warnings, numpy as np
warnings.simplefilter("ignore")
sys.path[:0] = [%(pkg)r, %(fake)r, %(root)r]
import vaex
from vaex_amd import vaex_filter as vf
n = 4000
r0 = np.random.default_rng(1)
df = vaex.from_arrays(x=r0.normal(0, 1, n), y=r0.normal(0, 1, n), i=r0.integers(-5, 5, n), idx=np.arange(n))
exprs = ["x > 0", "y < 0.5", "i != 2", "x + y > 1", "(x > -1) & (y > -1)", "i >= 0"]
expressible = 0
for seed in range(500):
    rng = np.random.default_rng(seed)
    d, steps = df, []
    for k in range(int(rng.integers(1, 5))):
        op = str(rng.choice(["getitem", "getitem", "filter_and", "filter_or", "filter_replace", "drop"]))
        e = str(rng.choice(exprs))
        if op == "getitem":
            d = d[d._expr(e)]; steps.append(("[]", e))
        elif op == "drop":
            d = d.drop_filter(); steps.append("drop")
        else:
            d = d.filter(e, mode=op.split("_")[1]); steps.append((op, e))
    expr = vf.filter_expression(d)
    if not d.filtered:
        assert expr is None, (steps, expr)
        continue
    if expr is None:     # (an "or" / "replace" link in the chain: the executor's host mask)
        continue
    expressible += 1
    kept = np.isin(np.arange(n), d.idx.to_numpy())
    assert np.array_equal(np.asarray(df.evaluate(expr)).astype(bool), kept), (seed, steps, expr)
assert expressible > 200, expressible
print("CHAINS OK", expressible)
"""


def test_random_filter_chains_resolve_to_the_rows_vaex_keeps():
    """df[a][b], df.filter(..., mode=and / or / replace), drop_filter in random order: the ONE expression a filtered frame's filter is compiled
    from (the device predicate in every aggregator's keep-mask) keeps exactly the rows vaex's own filter keeps — or is declared not expressible"""
    if not os.path.isdir(os.path.join(PKG, "vaex")):
        pytest.skip("oracle/_ref/vaexpy not built (run __graft_entry__.build() where /root/reference exists)")
    r = subprocess.run([sys.executable, "-c", CHAIN_FUZZ % dict(pkg=PKG, fake=FAKE, root=ROOT)], capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0 and "CHAINS OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
