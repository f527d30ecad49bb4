"""NAMED selections of the REAL vaex (df.select(...); df.count(selection=True) — vaex/dataframe.py:5041, vaex/selections.py:40-160)
under vaex_amd.install(): a name whose history resolves to an expression of the device predicate subset is planned like an
expression selection (vaex_amd/vaex_selection.py) — every mode of df.select (replace / and / or / subtract / xor), select_inverse,
undo / redo, a second name, next to a filter, in one task with other kinds of selections, delayed.  Results must be those of vaex's own
host evaluation, computed in the same process before install().

  * here (no GPU): the HIP classes are switched off, every task part falls back to vaex's own C++ and the planned predicate is evaluated
    with numpy in `process` — this pins the host logic (resolution of the history, the task's bookkeeping, the re-definition guard);
  * `-m gpu`: the product — the predicate runs on the device (vxh_agg_set_selection)."""
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
from vaex_amd import vaex_selection as vsel
n = %(n)d
def make():
    rng = np.random.default_rng(5)
    v = rng.normal(3, 2, n); v[::991] = np.nan
    return vaex.from_arrays(x=rng.normal(0, 1, n), y=rng.normal(0, 1, n), v=v, i=rng.integers(-20, 20, n).astype("i4"),
                            m=np.ma.array(rng.normal(0, 1, n), mask=rng.random(n) < 0.05))
L = [-4, 4]

# every scenario: (how the selections are made, which of its calls are expected on the device road)
def s_replace(d):
    d.select("x > 0.5")
    d.select("y < 0.25")                                   # replace: only this one counts
    return [d.count(binby="x", limits=L, shape=32, selection=True), d.mean("v", binby="y", limits=L, shape=16, selection="default")]
def s_and_or(d):
    d.select("x > -1"); d.select("y < 1", mode="and"); d.select("i > 15", mode="or")
    return [d.count(binby="x", limits=L, shape=32, selection=True), d.sum("v", binby="y", limits=L, shape=16, selection=True)]
def s_subtract(d):
    d.select("x > -1"); d.select("v > 4", mode="subtract")
    return [d.count(binby=["x", "y"], limits=[L, L], shape=8, selection=True), d.std("v", binby="x", limits=L, shape=8, selection=True)]
def s_inverse(d):
    d.select("(x > 0) & (y > 0)"); d.select_inverse()
    return [d.count(binby="x", limits=L, shape=32, selection=True), d.min("v", binby="x", limits=L, shape=8, selection=True)]
def s_undo_redo(d):
    d.select("x > 1"); d.select("x < -1"); d.selection_undo()
    a = d.count(binby="x", limits=L, shape=32, selection=True)
    d.selection_redo()
    return [a, d.count(binby="x", limits=L, shape=32, selection=True)]
def s_two_names(d):
    d.select("x > 0", name="pos"); d.select("i < 0", name="neg"); d.select("y > 1", mode="and", name="pos")
    return [d.count(binby="y", limits=L, shape=16, selection="pos"), d.sum("i", binby="y", limits=L, shape=16, selection="neg"),
            d.count(binby="y", limits=L, shape=16, selection=["pos", None, "neg", "v > 3"])]   # (a list: vaex's own masks)
def s_xor(d):
    d.select("x > 0"); d.select("y > 0", mode="xor")        # no device form: vaex's host masks
    return [d.count(binby="x", limits=L, shape=32, selection=True)]
def s_outside_subset(d):
    d.select("sin(x) + y ** 3 < 0.5")                      # libm functions / other powers: host
    a = d.count(binby="x", limits=L, shape=32, selection=True)
    d.select("m > 0")                                      # a column with missing values: host
    return [a, d.count(binby="x", limits=L, shape=32, selection=True)]
def s_filtered(d):
    f = d[d.y > -0.5]
    f.select("x > 0"); f.select("v < 5", mode="and")
    return [f.count(binby="x", limits=L, shape=32, selection=True), f.mean("v", binby="x", limits=L, shape=8, selection=True), f.count(selection=True)]
def s_one_task(d):                                         # a device name, an expression, nothing and a host name in ONE task
    d.select("x > 0"); d.select("sin(x) ** 2 < 0.5", name="host")
    a = d.count(binby="x", limits=L, shape=16, selection=True, delay=True)
    b = d.count(binby="x", limits=L, shape=16, selection="y > 0", delay=True)
    c = d.count(binby="x", limits=L, shape=16, delay=True)
    e = d.count(binby="x", limits=L, shape=16, selection="host", delay=True)
    g = d.sum("v", binby="x", limits=L, shape=16, selection=True, delay=True)
    d.execute()
    return [a.get(), b.get(), c.get(), e.get(), g.get()]
def s_arithmetic(d):                                       # round 5: arithmetic over float64 columns is evaluated on the device
    d.select("x ** 2 + y ** 2 < 2")
    a = d.count(binby="x", limits=L, shape=32, selection=True)
    d.select("(2 * x - y / 3 > -0.5) & (abs(v - 3) <= 1.5)")
    return [a, d.count(binby="x", limits=L, shape=32, selection=True), d.mean("v", binby="y", limits=L, shape=16, selection=True)]
def s_virtual(d):                                          # ... and so is a selection over a VIRTUAL column whose expression is
    d["r"] = (d.x ** 2 + d.y ** 2) ** 0.5                  # (a power other than 2: host)
    d["r2"] = d.x ** 2 + d.y ** 2
    d["rs"] = np.sqrt(d.x ** 2 + d.y ** 2)
    a = d.count(binby="x", limits=L, shape=32, selection="r2 < 2")
    b = d.count(binby="x", limits=L, shape=32, selection="(rs < 1.5) & (v > 2)")
    d.select("rs >= 0.5")
    c = d.sum("v", binby="y", limits=L, shape=16, selection=True)
    e = d.count(binby="x", limits=L, shape=32, selection="r < 1")
    return [a, b, c, e]
def s_zero_d(d):
    d.select("(i >= 3) | (x < -2)")
    return [np.array([float(d.count(selection=True)), float(d.sum("v", selection=True)), float(d.max("i", selection=True))])]
scenarios = dict(replace=(s_replace, 2), and_or=(s_and_or, 2), subtract=(s_subtract, 2), inverse=(s_inverse, 2), undo_redo=(s_undo_redo, 2), two_names=(s_two_names, 2),
                 xor=(s_xor, 0), outside_subset=(s_outside_subset, 0), filtered=(s_filtered, 3), one_task=(s_one_task, 3), zero_d=(s_zero_d, 3),
                 arithmetic=(s_arithmetic, 3), virtual=(s_virtual, 3))

want = {name: fn(make()) for name, (fn, _) in scenarios.items()}      # plain vaex, its own C++, its own host masks

# the resolution of a history, by itself
d = make()
d.select("x > 0"); d.select("y > 0", mode="and"); d.select("i == 3", mode="or"); d.select("v > 4", mode="subtract")
assert vsel.named_expression(d, "default") == "((((x > 0)) & (y > 0)) | (i == 3)) & ~(v > 4)", vsel.named_expression(d, "default")
d.select_inverse()
assert vsel.named_expression(d, "default").startswith("~(") and vsel.named_expression(d, "nope") is None
d.select("y > 0", mode="xor")
assert vsel.named_expression(d, "default") is None
d.select_lasso("x", "y", [0, 1, 1], [0, 0, 1])
assert vsel.named_expression(d, "default") is None
d.select_nothing()
assert vsel.named_expression(d, "default") is None

# ... and against vaex's own evaluation of the name, over random histories (modes, inversions, undo / redo; <= 4 distinct comparisons so
# that most of them stay inside the device subset): the rows the resolved predicate keeps are the rows vaex's host mask keeps
from vaex_amd import predicate as vpred
d = make()
host = {c: d[c].to_numpy() for c in ("x", "y", "v", "i")}
known = {c: host[c] for c in host}
rng = np.random.default_rng(77)
terms = ["x > 0.3", "y <= -0.2", "v >= 3.5", "i != 4", "x < 1.5", "i < 0"]
checked = unsupported = 0
for trial in range(60):
    d.select_nothing()
    pool = list(rng.choice(terms, 3, replace=False))
    for step in range(int(rng.integers(1, 6))):
        what = rng.random()
        if what < 0.12 and d.has_selection():
            d.select_inverse()
        elif what < 0.2 and d.has_selection() and d.selection_can_undo():
            d.selection_undo()
        elif what < 0.25 and d.selection_can_redo():
            d.selection_redo()
        else:
            e = str(rng.choice(pool))
            if rng.random() < 0.3:
                e = f"({e}) & ({rng.choice(pool)})"
            d.select(e, mode=str(rng.choice(["replace", "and", "or", "subtract", "and", "or", "xor"])))
    expr = vsel.named_expression(d, "default")
    if expr is None:
        unsupported += 1
        continue
    try:
        pred = vpred.compile_selection(expr, known)
    except vpred.Unsupported:
        unsupported += 1
        continue
    mine = pred.numpy_mask(host)
    theirs = d.count(binby="x", limits=L, shape=64, selection=True)
    xs = host["x"][mine]
    assert np.array_equal(theirs, np.histogram(xs[xs == xs], bins=64, range=L)[0]), (trial, expr)
    checked += 1
assert checked >= 30, (checked, unsupported)
print("ok fuzz", checked, "histories resolved and equal to vaex's masks,", unsupported, "left to vaex", flush=True)

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
def _recording_decode(cls, encoding, spec, df, nthreads):
    part = _decode(cls, encoding, spec, df, nthreads)
    used.append((part.backend_used, dict(spec.get(vsel.SPEC_KEY) or {}), len(part._hip_selections)))
    return part
_task.decode = classmethod(_recording_decode)

for name, (fn, planned_named) in scenarios.items():
    before = dict(vsel.stats); del used[:]
    got = fn(make())
    delta = {k: vsel.stats[k] - before[k] for k in before}
    assert len(got) == len(want[name]), name
    for k, (g, w) in enumerate(zip(got, want[name])):
        g, w = np.asarray(g, dtype="f8"), np.asarray(w, dtype="f8")
        assert g.shape == w.shape and np.array_equal(np.isnan(g), np.isnan(w)), (name, k)
        exact = np.array_equal(np.nan_to_num(g), np.nan_to_num(w))
        assert exact or np.all(np.abs(np.nan_to_num(g) - np.nan_to_num(w)) <= 1e-12 * 10.0 * n), (name, k, np.nanmax(np.abs(g - w)))
        if np.all(np.nan_to_num(w) == np.round(np.nan_to_num(w))) and k == 0:
            assert exact, (name, k)   # counts
    if planned_named == 0:
        assert all(not u[1] for u in used), (name, used)
    else:
        assert any(u[1] for u in used), (name, used)
        assert delta["planned"] >= planned_named, (name, delta)
        assert delta["device_chunks" if GPU else "host_chunks"] > 0, (name, delta)
        assert all(u[0] == ("hip" if GPU else "cpu") for u in used), (name, used)
    print("ok", name, delta, used[:2], flush=True)

# a name re-defined between scheduling (delay=True) and execution: vaex merges delayed aggregations into their task when the run starts
# (vaex/execution.py:_merge) and looks the name up then — the NEW definition counts, here too (the name is resolved in that merge)
d = make()
d.select("x > 0")
a = d.count(binby="x", limits=L, shape=8, selection=True, delay=True)
d.select("x < 0")
d.execute()
xs = d.x.to_numpy()
assert np.array_equal(a.get(), np.histogram(xs[xs < 0], bins=8, range=L)[0]), a.get()
print("ok re-defined before the run")
# ... and should a name ever change between that merge and the task part's decode, the part refuses (the planned predicate would be the
# old definition, vaex's mask the new one)
class _Desc:
    name = "AggCount"; selection = "default"; expressions = []
class _Part:
    df = d; aggregation_descriptions = [_Desc()]; aggregations = []
try:
    vsel.attach(_Part(), "cpu", None, 2, named={"0": "(x > 0)"})
    raise SystemExit("the re-defined selection went unnoticed")
except RuntimeError as e:
    assert "re-defined" in str(e), e
vsel.attach(_Part(), "cpu", None, 2, named={"0": "(x < 0)"})
print("ok re-definition guard")
got = d.count(binby="x", limits=L, shape=8, selection=True)
vaex_amd.uninstall()
import vaex.tasks
assert vaex.tasks.TaskAggregations.encode.__qualname__.startswith("TaskAggregations.") and vaex.tasks.TaskAggregations.add_aggregation_operation.__qualname__.startswith("TaskAggregations.")
assert np.array_equal(got, d.count(binby="x", limits=L, shape=8, selection=True))
print("NAMED OK", flush=True)
'''


def _run(gpu, n):
    if not os.path.isdir(os.path.join(PKG, "vaex")):
        pytest.skip("oracle/_ref/vaexpy not built (run __graft_entry__.build() where /root/reference exists)")
    env = dict(os.environ, VAEX_NUM_THREADS=os.environ.get("VAEX_NUM_THREADS", "4"))
    env.setdefault("VAEX_HOME", "/tmp/vaex_home_named")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(pkg=PKG, fake=FAKE, root=ROOT, gpu=gpu, n=n)], cwd="/tmp", capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "NAMED OK" in r.stdout, (r.stdout[-3000:], r.stderr[-4000:])
    return r.stdout


def test_named_selections_host_logic_on_vaex_cpp():
    _run(0, 200_000)


@pytest.mark.gpu
def test_named_selections_on_the_device():
    out = _run(1, 2_000_000)
    assert out.count("\nok") + out.startswith("ok") >= 12, out


HISTORY_FUZZ = r"""
import sys, warnings, numpy as np
warnings.simplefilter("ignore")
sys.path[:0] = [%(pkg)r, %(fake)r, %(root)r]
import vaex
from vaex_amd import vaex_selection as vs
n = 4000
r0 = np.random.default_rng(1)
df = vaex.from_arrays(x=r0.normal(0, 1, n), y=r0.normal(0, 1, n), i=r0.integers(-5, 5, n), idx=np.arange(n))
exprs = ["x > 0", "y < 0.5", "i != 2", "x + y > 1", "(x > -1) & (y > -1)", "i >= 0"]
expressible = 0
for seed in range(400):
    rng = np.random.default_rng(seed)
    d = df.copy()
    name = str(rng.choice(["default", "other"]))
    steps = []
    for k in range(int(rng.integers(1, 7))):
        op = str(rng.choice(["select", "select", "select", "inverse", "nothing", "undo", "redo"]))
        if op == "select":
            e, mode = str(rng.choice(exprs)), str(rng.choice(["replace", "and", "or", "subtract", "xor"]))
            d.select(e, mode=mode, name=name); steps.append((e, mode))
        elif op == "inverse":
            d.select_inverse(name=name); steps.append(op)
        elif op == "nothing":
            d.select_nothing(name=name); steps.append(op)
        elif op == "undo" and d.selection_can_undo(name):
            d.selection_undo(name); steps.append(op)
        elif op == "redo" and d.selection_can_redo(name):
            d.selection_redo(name); steps.append(op)
    expr = vs.named_expression(d, name)
    if expr is None:        # (xor in the history, or nothing selected: vaex's own masks)
        continue
    expressible += 1
    want = np.isin(np.arange(n), d.evaluate("idx", selection=name))      # the rows vaex's selection machinery keeps
    got = np.asarray(d.evaluate(expr)).astype(bool)
    assert np.array_equal(got, want), (seed, steps, expr, int(got.sum()), int(want.sum()))
assert expressible > 150, expressible
print("HISTORIES OK", expressible)
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_random_selection_histories_resolve_to_the_rows_vaex_keeps():
    """a named selection's history (select with every mode, inverse, nothing, undo, redo, in random order) as ONE expression — what the device
    predicate of `selection=True` / `selection="name"` is compiled from — keeps exactly the rows vaex's own selection machinery keeps"""
    r = subprocess.run([sys.executable, "-c", HISTORY_FUZZ % dict(pkg=PKG, fake=FAKE, root=ROOT)], capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0 and "HISTORIES OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
