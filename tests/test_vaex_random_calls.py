"""Randomised differential test through the REAL vaex API (the fixed list of calls is tests/test_vaex_differential.py): 1200 calls drawn from
a grammar — statistic x expression (plain / float32 / integer / bool / big-endian / masked / virtual / arithmetic) x 0-3 binby dimensions
(float, integer, virtual, big-endian, masked columns; fixed or data-derived limits; random shapes) x selection (none / a random expression of
the predicate grammar / a list / a named selection) x frame (plain / filtered inside and outside the device-predicate subset / sliced /
both) x immediate or delayed in batches — run once under vaex_amd.install() and once on vaex's own C++ after uninstall(), compared call by
call: integers exactly, fp64 sums / means to 1e-12 of the result's magnitude, variances to the cancellation bound, and an exception on one
side must be the same exception on the other.  Without a GPU the first half runs install()'s host logic alone (task parts, keep-mask filters, planned selections as host masks) over vaex's C++."""
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
import vaex
from tests.predicate_fuzz import random_expression
gpu = %(gpu)d
ncalls = %(ncalls)d
big = %(big)d     # 1: millions of rows and grids beyond one workgroup's LDS — the partition passes, the hot box, the selections evaluated inside them
n = 4_000_000 if big else 120_000
def make():
    r = np.random.default_rng(21)
    x = r.normal(0, 1, n); x[::997] = np.nan
    v = r.normal(3, 2, n); v[::501] = np.nan
    cols = dict(x=x, y=r.normal(0, 1, n), z=r.normal(0, 2, n), v=v, f4=r.normal(0, 1, n).astype("f4"),
                i=r.integers(-50, 50, n).astype("i4"), h=r.integers(-300, 300, n).astype("i2"), u1=r.integers(0, 200, n).astype("u1"), b=r.random(n) < 0.3,
                k=r.integers(0, 30, n), be=r.normal(0, 1, n).astype(">f8"), m=np.ma.array(r.normal(0, 1, n), mask=r.random(n) < 0.05))
    df = vaex.from_arrays(**cols)
    df["r"] = np.sqrt(df.x ** 2 + df.y ** 2)
    df["vi"] = df.v * df.i
    return df
LIMITS = dict(x=[-3, 3], y=[-3, 3], z=[-5, 5], f4=[-2.5, 2.5], i=[-50.5, 49.5], h=[-100, 100], u1=[0, 256], k=[-0.5, 29.5], be=[-3, 3], m=[-2, 2], r=[0, 4], v=[-3, 9])
VALUES = ["x", "y", "v", "f4", "i", "h", "u1", "be", "m", "r", "vi", "x*2+y", "b"]
SCALE = {"x": 5, "y": 5, "v": 12, "f4": 5, "i": 50, "h": 300, "u1": 200, "be": 5, "m": 5, "r": 7, "vi": 600, "x*2+y": 15, "b": 1}   # largest magnitudes of the values
STATS = ["count", "count", "sum", "mean", "mean", "std", "var", "min", "max", "minmax", "count_star", "nunique"]   # (nunique: the distinct-key task on the device hash set, also batched with the others)
SEL_COLS = ["x", "y", "v", "f4", "i", "h", "u1", "b"]
def frames(df):
    return {"plain": df, "filtered": df[df.x > -0.5], "filtered_libm": df[np.sin(df.y * 3) > -0.5], "sliced": df[3000:n - 10_000], "both": df[df.v < 5][1000:n - 30_000]}
def draw(seed):
    r = np.random.default_rng(5000 + seed)
    c = dict(frame=str(r.choice(["plain", "plain", "filtered", "filtered_libm", "sliced", "both"])), stat=str(r.choice(STATS)), value=str(r.choice(VALUES)))
    nd = int(r.choice([0, 1, 1, 2, 2, 3]))
    c["binby"] = [str(b) for b in r.choice(list(LIMITS), size=nd, replace=False)]
    c["shape"] = [int(r.integers(1, 70 if nd < 3 else 20)) for _ in range(nd)]
    if big and nd:
        c["shape"] = [int(r.choice([128, 256, 300, 512])) for _ in range(nd)] if nd < 3 else [int(r.choice([48, 64, 128])) for _ in range(nd)]
    c["limits"] = "minmax" if (nd and r.random() < 0.12 and not set(c["binby"]) & {"m"}) else [LIMITS[b] for b in c["binby"]]
    s = r.random()
    def sel():
        for t in range(20):
            e = random_expression(np.random.default_rng(int(r.integers(1 << 30))), SEL_COLS, ["x", "y", "v"])
            if len(e) < 160:
                return e
        return "x > 0"
    c["selection"] = None if s < 0.4 else (sel() if s < 0.75 else ([None, sel()] if s < 0.87 else "named"))
    c["named"] = sel() if c["selection"] == "named" else None
    c["delayed"] = bool(r.random() < 0.25)
    if c["stat"] in ("min", "max", "minmax") and c["value"] == "b":
        c["value"] = "u1"
    if c["stat"] == "minmax":
        c["binby"], c["shape"], c["limits"] = [], [], []
    if c["stat"] == "nunique":
        c["binby"], c["shape"], c["limits"] = [], [], []
        c["value"] = str(r.choice(["i", "h", "u1", "k", "f4", "b"]))
        if isinstance(c["selection"], list):
            c["selection"] = c["selection"][1]
    return c
def call(d, c, delay=False):
    kw = {}
    if c["binby"]:
        kw.update(binby=c["binby"], limits=c["limits"], shape=c["shape"])
    sel = c["selection"]
    if sel == "named":
        d.select(c["named"], name="fuzz")
        sel = "fuzz"
    if sel is not None:
        kw["selection"] = sel
    if delay:
        kw["delay"] = True
    if c["stat"] == "count_star":
        return d.count(**kw)
    if c["stat"] == "nunique":
        return d[c["value"]].nunique(**kw)
    return getattr(d, c["stat"])(c["value"], **kw)
import os
_only = os.environ.get("VAEX_AMD_RANDOM_CALL_RANGE")   # (a diagnosis run: "lo:hi" — these calls only, what execute() raised printed)
call_range = range(*[int(q) for q in _only.split(":")]) if _only else range(ncalls)
def run_all(tag):
    df = make()
    fr = frames(df)
    out, pending = {}, []
    def flush():
        if pending:
            try:
                fr["plain"].execute()
            except Exception as e:
                if _only:
                    import traceback
                    print(tag, "execute() raised with", [i for i, p in pending], "pending:", "".join(traceback.format_exception(type(e), e, e.__traceback__))[-2500:])
            for i, p in pending:
                try:
                    out[i] = p.get()
                except Exception as e:
                    out[i] = ("EXC", type(e).__name__, str(e)[:160])
            del pending[:]
    for i in call_range:
        c = draw(i)
        d = fr[c["frame"]]
        try:
            if c["delayed"]:
                pending.append((i, call(d, c, delay=True)))
                if len(pending) >= 3:
                    flush()
            else:
                flush()
                out[i] = call(d, c)
        except Exception as e:
            flush()
            out[i] = ("EXC", type(e).__name__, str(e)[:160])
    flush()
    return out
def flat(v):
    if isinstance(v, tuple) and v and v[0] == "EXC":
        return [("EXC", v[1], v[2])]
    if isinstance(v, (list, tuple)):
        r = []
        for p in v:
            r += flat(p)
        return r
    a = np.ma.asarray(v)
    return [np.ma.filled(a.astype("f8"), np.nan)]
import vaex_amd
from vaex_amd import vaex_selection as vsel, vaex_filter as vflt
if gpu:
    assert vaex_amd.superagg.device_count() > 0
    vaex_amd.install()
else:
    # without a GPU: install()'s HOST logic alone — the task parts, the filtered runs in the keep-mask form, the selections planned and then
    # evaluated as host masks — over vaex's own C++ classes (the HIP classes switched off, as tests/test_vaex_filter.py does)
    backend = vaex_amd.install(hash_sets=False, legacy=False, groupby=False)
    class _NoHip:
        def __getattr__(self, name):
            raise NotImplementedError("test: HIP classes switched off")
    backend.__dict__["_hip"] = _NoHip()
first = run_all("hip" if gpu else "cpu-1")
if not gpu:
    print("host logic alone: filtered runs in the keep-mask form:", vflt.stats["runs_switched"], "left pre-filtered:", vflt.stats["runs_mixed"], "| selections planned:", vsel.stats["planned"])
    assert vflt.stats["runs_switched"] > 5 or _only
    vaex_amd.uninstall()
if gpu:
    print("task parts on the HIP classes:", vaex_amd.task_stats["hip"], "on vaex's C++:", vaex_amd.task_stats["cpu"], vaex_amd.task_stats["cpu_reasons"])
    print("selections as device predicates (chunks):", vsel.stats["device_chunks"], "host masks:", vsel.stats["host_chunks"], "| filtered runs in the keep-mask form:", vflt.stats["runs_switched"], "left pre-filtered:", vflt.stats["runs_mixed"])
    assert (vaex_amd.task_stats["hip"] > 10 * max(1, vaex_amd.task_stats["cpu"]) and vsel.stats["device_chunks"] > 20) or _only
    # what the HIP entry answers where the reference raises (KNOWN_DEFECT below): the extrema of the rows the selection keeps
    dfp = make()
    keep = (dfp.x.to_numpy() > 2.5) & (dfp.y.to_numpy() > 2.5)          # a handful of rows: most chunks have none
    assert 0 < keep.sum() < 1500, keep.sum()
    got = dfp.minmax("v", selection="(x > 2.5) & (y > 2.5)")
    vs = dfp.v.to_numpy()[keep]
    assert np.array_equal(np.asarray(got), np.array([np.nanmin(vs), np.nanmax(vs)])), (got, vs)
    print("minmax(v, selection keeping", int(keep.sum()), "rows) under install():", np.asarray(got), "| of a selection keeping none:", np.asarray(dfp.minmax("v", selection="x > 100")))
    # a delayed minmax (the legacy statistic task: vaexfast.statisticNd from the pool's threads) in the SAME pass as a binned aggregation: the
    # two must not share a thread slot of the library (round 5: the legacy entry moved to an auxiliary slot; before, one of the two results
    # was corrupted in ~1 %% of such passes — this test's soak run)
    mixed = []
    for rep in range(250):
        a = dfp.count(binby="x", limits=[-3, 3], shape=32, delay=True)
        b = dfp.minmax("v", delay=True)
        c3 = dfp.sum("y", binby=["x", "z"], limits=[[-3, 3], [-5, 5]], shape=[9, 7], delay=True)
        dfp.execute()
        mixed.append((np.asarray(a.get()), np.asarray(b.get()), np.asarray(c3.get())))
    vaex_amd.uninstall()
    a, b, c3 = dfp.count(binby="x", limits=[-3, 3], shape=32), dfp.minmax("v"), dfp.sum("y", binby=["x", "z"], limits=[[-3, 3], [-5, 5]], shape=[9, 7])
    wrong = [rep for rep, (pa, pb, pc) in enumerate(mixed) if not (np.array_equal(pa, a) and np.array_equal(pb, b) and np.allclose(pc, c3, rtol=1e-12, atol=1e-9))]
    assert not wrong, ("a delayed minmax and binned aggregations in one pass", wrong[:10], len(wrong))
    print("250 passes holding a delayed minmax next to two binned aggregations: all equal to the reference")
second = run_all("cpu")
# A defect of the reference this test keeps finding: its legacy statistic task (df.minmax, limits="minmax": vaex/cpu.py:488-623) hands
# vaexfast.statisticNd the selected rows of every chunk, and a chunk in which the selection keeps NO row is an empty array whose stride the C
# wrapper refuses (src/vaexfast.cpp:160-195) — `df.minmax("y", selection="x == 4")` raises ValueError in plain vaex.  The HIP entry takes the
# empty chunk for what it is.  Such calls are counted, not compared.
KNOWN_DEFECT = "object_to_numpy1d_nocopy_endian: stride is not equal to 1"
bad, excs, known = [], 0, 0
# ... and its collateral (call 5332 of a soak run): where the reference's minmax raised, the HIP entry's answer goes on into vaex's own next step — with a LIST of
# selections and limits="minmax" that is a BinnerScalar whose limits are arrays, which vaex cannot hash when it merges the pass's tasks (TypeError in
# vaex/execution.py _merge: the same call would fail there on the reference too, had its minmax not raised first) — and the failed execute() leaves the OTHER
# delayed calls of the same batch pending on the HIP side.  A pending promise next to such a call is counted with it.
_force = os.environ.get("VAEX_AMD_RANDOM_FORCE_MOVED")   # (a check of the tie-break below itself: the reference's two answers of this call are spoilt on purpose)
if _force and not isinstance(second[int(_force)], tuple):
    second[int(_force)] = np.asarray(second[int(_force)], dtype="f8") + 1.0
defect_at = {i for i in call_range if any(isinstance(q, tuple) and KNOWN_DEFECT in q[2] for q in flat(second[i]))}
for i in call_range:
    c = draw(i)
    a, b = flat(first[i]), flat(second[i])
    if len(a) != len(b):
        bad.append((i, c, "different structure")); continue
    for p, q in zip(a, b):
        if isinstance(p, tuple) or isinstance(q, tuple):
            excs += 1
            if isinstance(q, tuple) and KNOWN_DEFECT in q[2]:   # (whatever the HIP side then ran into further on)
                known += 1      # (the reference raises, the HIP entry answers: see KNOWN_DEFECT)
                continue
            if isinstance(p, tuple) and "promise is still pending" in p[2] and not isinstance(q, tuple) and c["delayed"] and any(j in defect_at for j in range(i - 2, i + 3)):
                known += 1      # (a batch neighbour of such a call)
                continue
            if not (isinstance(p, tuple) and isinstance(q, tuple) and p[:2] == q[:2]):
                bad.append((i, c, "exception on one side only / another exception", first[i] if isinstance(p, tuple) else "result", second[i] if isinstance(q, tuple) else "result"))
            continue
        if p.shape != q.shape:
            bad.append((i, c, "shape", p.shape, q.shape)); continue
        if c["stat"] in ("std", "var"):
            # a cell with ONE row (or equal rows) has variance 0 +- an ulp of x^2: the reference sums pow(x, 2) (glibc: within an ulp of x * x, not
            # always equal to it — src/agg_sum.cpp:159) and subtracts numpy's mean * mean, so its variance of such a cell is now and then
            # -1e-17 and its std NaN, where x * x on the device gives 0 exactly.  Noise-level values count as equal to a NaN of that origin.
            noise = np.isnan(p) != np.isnan(q)
            tiny = 1e-7 * SCALE[c["value"]] if c["stat"] == "std" else 1e-13 * SCALE[c["value"]] ** 2
            if not np.all(np.abs(np.where(np.isnan(p), q, p)[noise]) <= tiny * max(1.0, float(np.nanmax(np.abs(q))) if np.isfinite(q).any() else 1.0)):
                bad.append((i, c, "NaN pattern beyond rounding noise", int(noise.sum()))); continue
            p, q = np.where(noise, 0.0, p), np.where(noise, 0.0, q)
        if not np.array_equal(np.isnan(p), np.isnan(q)):
            bad.append((i, c, "NaN pattern", int((np.isnan(p) != np.isnan(q)).sum()))); continue
        if c["stat"] in ("std", "var"):
            # (a variance is a difference of two moments of size mean^2: +- 1e-16 x mean^2 of rounding noise, whose square root — up to 1e-7 for the
            #  values of these columns — is what the std of a cell with one row, or equal rows, comes out as on either side)
            mag = SCALE[c["value"]]     # (noise of the variance: ~1e-16 x the second moment; of the std: its square root)
            ok = np.allclose(p, q, rtol=1e-7, atol=1e-7 * mag if c["stat"] == "std" else 1e-13 * mag * mag, equal_nan=True)
        elif c["stat"] in ("count", "count_star", "min", "max", "minmax", "nunique"):
            ok = np.array_equal(p, q, equal_nan=True)
        else:
            fin = np.abs(q[np.isfinite(q)])
            scale = max(float(fin.max()) if fin.size else 0.0, 1.0) * (n if c["stat"] == "sum" and False else 1.0)
            ok = np.allclose(p, q, rtol=1e-11, atol=1e-11 * scale, equal_nan=True)
        if not ok:
            with np.errstate(invalid="ignore"):
                bad.append((i, c, "values", float(np.nanmax(np.abs(p - q)))))
if bad and gpu:
    # which side moved?  Every differing call once more on both sides, alone (round 6: one intermittent difference per ~3 full-suite runs — a variance of
    # the masked column with a selection on a filtered frame — that no repeat of the call alone reproduces: tools/r07_var_masked_stress.py)
    fr2 = frames(make())
    again_ref = {b[0]: flat(call(fr2[draw(b[0])["frame"]], draw(b[0]))) for b in bad if not draw(b[0])["delayed"]}
    if _force and int(_force) in again_ref:
        again_ref[int(_force)] = [q + 2.0 for q in again_ref[int(_force)]]
    vaex_amd.install()
    again_hip = {i: flat(call(fr2[draw(i)["frame"]], draw(i))) for i in again_ref}
    vaex_amd.uninstall()
    moved = 0
    def reference_on_one_thread(i):
        # the tie-break when the reference's two answers differ from each other AND from the (stable) HIP answer — seen on the 4e6-row form of this test, call 50, a
        # variance of the masked column: the reference once more with ONE pool thread, where `grid_used` (src/agg_base.hpp:17, a vector<bool> the pool's threads write
        # concurrently) cannot lose a thread's grid
        import vaex.execution, vaex.multithreading
        ex1 = vaex.execution.ExecutorLocal(vaex.multithreading.ThreadPoolIndex(max_workers=1))
        fr1 = frames(make())
        for f in fr1.values():
            f.executor = ex1
        return flat(call(fr1[draw(i)["frame"]], draw(i)))
    for i in again_ref:
        eq = lambda u, w: all(isinstance(p, np.ndarray) and isinstance(q, np.ndarray) and p.shape == q.shape and np.allclose(p, q, rtol=1e-9, atol=1e-12, equal_nan=True) for p, q in zip(u, w))
        hip_stable, ref_stable, agree_now = eq(flat(first[i]), again_hip[i]), eq(flat(second[i]), again_ref[i]), eq(again_hip[i], again_ref[i])
        print("AGAIN", i, "| first HIP run == reference:", eq(flat(first[i]), flat(second[i])), "| HIP again == reference again:", agree_now,
              "| first HIP run == HIP again:", hip_stable, "| reference == reference again:", ref_stable)
        if hip_stable and agree_now and not ref_stable:
            # the REFERENCE's first answer was the outlier: its second run, alone, equals both HIP runs.  Caught on the GPU box (profiles/r06_reference_moved.txt:
            # call 946, a mean of the masked column — the reference's per-thread grids race, INTEGRATION.md "Differences": grid_used is a vector<bool>
            # written by concurrent threads).  Counted, not compared.
            moved += 1
            bad = [b for b in bad if b[0] != i]
        elif hip_stable and not ref_stable:
            alone = reference_on_one_thread(i)
            print("AGAIN", i, "| the reference on ONE pool thread == both HIP runs:", eq(again_hip[i], alone), "| == its own first / second answer:", eq(flat(second[i]), alone), eq(again_ref[i], alone))
            if eq(again_hip[i], alone):
                moved += 1
                bad = [b for b in bad if b[0] != i]
    if moved:
        print("the reference's own answer moved between two runs (the HIP answer did not):", moved)
print("calls", ncalls, "of which raised on both sides alike:", excs - known, "| the reference raised on an empty selected chunk, the HIP entry answered:", known, "| different:", len(bad))
for bline in bad[:12]:
    print("BAD", bline)
assert not bad
print("DONE")
'''


def _run(gpu, ncalls, timeout, big=0):
    env = dict(os.environ, VAEX_NUM_THREADS=os.environ.get("VAEX_NUM_THREADS", "4"))
    out = subprocess.run([sys.executable, "-c", SCRIPT % dict(pkg=PKG, fake=FAKE, root=ROOT, gpu=gpu, ncalls=ncalls, big=big)], cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and "DONE" in out.stdout, out.stdout[-5000:] + out.stderr[-5000:]
    report = os.environ.get("VAEX_AMD_REPORT_DIR")
    if report and gpu:
        with open(os.path.join(report, "random_calls_big_report.txt" if big else "random_calls_report.txt"), "w") as f:
            f.write("\n".join(line for line in out.stdout.splitlines() if not line.startswith("BAD")))
    return out.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_random_calls_through_the_host_logic_alone_agree_with_the_reference():
    _run(0, 500, 900)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_random_calls_agree_with_the_reference():
    _run(1, int(os.environ.get("VAEX_AMD_RANDOM_CALLS", "1200")), 1500)   # (a soak run raises it)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_random_calls_on_millions_of_rows_and_large_grids_agree_with_the_reference():
    _run(1, int(os.environ.get("VAEX_AMD_RANDOM_CALLS_BIG", "90")), 2400, big=1)
