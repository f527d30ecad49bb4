"""Randomised differential test of df.groupby through the REAL vaex API: random key dtypes / ranges (dense, with gaps, wide, a single key,
negative, at the ends of the dtype), one to three keys, random sets of aggregations (with and without their own selections), sort options,
slices, filters, immediate and delayed — the wrapped groupby (device groupby, its task form for delay=True, or the decline to vaex's own passes
on the HIP classes) against the original.  Without a GPU the device groupby is stood in for by binned.Frame over the reference's own classes
(one key, no hash path), which checks the host logic: the plan, the key column's type, the sort order, the frame's assembly.
Three defects of the reference's own groupby are recognised and counted instead of compared (INTEGRATION.md "Differences", pinned both ways in
tests/test_vaex_groupby.py): the labels of a bool key sorted descending, key ranges that do not fit the key's own dtype, and the IndexError of
BinnerInteger on unsigned / single keys sorted descending."""
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
import sys, warnings, numpy as np
warnings.simplefilter("ignore")
sys.path[:0] = [%(pkg)r, %(fake)r, %(root)r]
import vaex, vaex_amd
from vaex_amd import vaex_groupby as vg, binned
gpu = %(gpu)d
ncalls = %(ncalls)d
if gpu:
    assert vaex_amd.superagg.device_count() > 0
    vaex_amd.install()
    original = vaex.dataframe.DataFrameLocal.groupby.__wrapped__
else:
    import threading
    from tests.test_golden_api import RefAdapter
    ref = RefAdapter(vaex.superagg)
    class HostMaskFrame(binned.Frame):
        def _selection_mask(self, selection):
            sel = super()._selection_mask(selection)
            if isinstance(sel, binned._predicate.Predicate):
                key = ("mask", sel.key())
                if key not in self._predicates:
                    self._predicates[key] = sel.numpy_mask({c: self.columns[c] for c in sel.columns})
                return self._predicates[key]
            return sel
    vg._frame_for = lambda df, columns: HostMaskFrame(dict(columns), chunk_size=7_000, nthreads=2, superagg=ref)
    class HostCollector:
        def __init__(self, plan, capacity):
            self.parts, self.lock, self.names, self.dtypes, self.rows = [], threading.Lock(), list(plan.columns), {k: a.dtype for k, a in plan.columns.items()}, 0
        def append(self, chunks):
            with self.lock:
                self.parts.append({k: np.array(v) for k, v in chunks.items()})
                self.rows += len(next(iter(chunks.values())))
        def frame(self):
            return HostMaskFrame({k: np.concatenate([p[k] for p in self.parts]) if self.parts else np.zeros(0, dtype=self.dtypes[k]) for k in self.names}, chunk_size=7_000, nthreads=2, superagg=ref)
    vg._collector_for = lambda plan, capacity: HostCollector(plan, capacity)
    state = {}
    vg.install(vaex, state)
    original = state["groupby"][1]
A = vaex.agg
def cols(d, keys):
    d = d.sort(keys)
    return {c: d[c].to_numpy() for c in d.get_column_names()}
def same(a, b):
    assert list(a) == list(b), ("columns", list(a), list(b))
    for c in a:
        x, y = a[c], b[c]
        assert len(x) == len(y), ("groups", c, len(x), len(y))
        assert np.ma.isMaskedArray(x) == np.ma.isMaskedArray(y), ("masked-ness", c, type(x), type(y))
        if np.ma.isMaskedArray(x):   # (the same entries are missing; what the arrays hold UNDER the mask is nobody's result)
            mx, my = np.ma.getmaskarray(x), np.ma.getmaskarray(y)
            assert np.array_equal(mx, my), ("mask", c, mx[:10], my[:10])
            x, y = np.ma.getdata(x).copy(), np.ma.getdata(y).copy()
            x[mx] = 0; y[my] = 0
        x, y = np.ma.getdata(x), np.ma.getdata(y)
        assert x.dtype == y.dtype, ("dtype", c, x.dtype, y.dtype)
        if x.dtype.kind in "OUS":   # (labels of a categorical key)
            assert x.tolist() == y.tolist(), (c, x[:10], y[:10])
        elif x.dtype.kind in "iub":
            assert np.array_equal(x, y), (c, x[:10], y[:10])
        else:
            assert np.allclose(x, y, rtol=1e-9, atol=1e-9, equal_nan=True), (c, x[:10], y[:10])
def key_column(rng, n, kdt, style):
    if kdt == "bool":
        return rng.integers(0, 2, n).astype(bool)
    info = np.iinfo(kdt)
    if style == "dense": k = rng.integers(0, min(40, info.max), n)
    elif style == "gappy": k = rng.integers(0, 12, n) * int(min(9, info.max // 12))
    elif style == "wide":
        k = rng.integers(info.min // 2, info.max // 2, n, dtype="i8") if info.bits < 64 else rng.integers(-2**40, 2**40, n)
        if n > 200: k = rng.choice(k[:int(rng.choice([20, 150, 3000]))], n)
    elif style == "single": k = np.full(n, min(7, info.max))
    elif style == "negative": k = rng.integers(max(info.min, -30), min(info.max, 5), n)
    else: k = rng.choice(np.array([info.min, info.min + 1, 0, info.max - 1, info.max], dtype="u8" if kdt == "u8" else "i8"), n)
    return np.asarray(k).astype(kdt)
bad, known, paths = [], {}, {}
import os
only = [int(q) for q in os.environ.get("VAEX_AMD_RANDOM_SEEDS", "").split(",") if q]   # (a diagnosis run: these seeds only, both results printed)
like_gpu = gpu or bool(os.environ.get("VAEX_AMD_RANDOM_GPU_RNG"))   # (a diagnosis run on the host stand-in: draw what the GPU run draws for a seed, so that a single-key call a soak run found can be replayed without a GPU)
for seed in (only or range(ncalls)):
    rng = np.random.default_rng(seed)
    vg.device_coding_min_rows = 1000 if (gpu and seed %% 2) else 2_000_000   # (every other call: keys / values with missing entries and float keys are coded on the device, vxh_code_column)
    n = int(rng.choice([1, 2, 3, 10, 100, 5000, 20000]))
    nkeys = int(rng.choice([1, 1, 1, 2, 3])) if like_gpu else 1
    if nkeys > 1 and not gpu:
        continue
    kinds = [str(rng.choice(["i1", "i2", "i4", "i8", "u1", "u2", "u4", "bool"])) for _ in range(nkeys)]
    styles = [str(rng.choice(["dense", "gappy", "wide", "single", "negative", "extreme"])) for _ in range(nkeys)]
    if not gpu and styles[0] in ("wide", "extreme") and kinds[0] not in ("i1", "u1", "i2", "u2", "bool"):
        continue      # (scattered keys need the hash aggregation: no stand-in without a GPU)
    data = {f"k{j}": key_column(rng, n, kinds[j], styles[j]) for j in range(nkeys)}
    # round 6 (late): keys that are not handed back as the integers the device grouped — missing values in a key (a numpy mask), a categorical key
    # (dense, non-negative codes), a float key with NaN (scattered bit patterns: the device only) — _finish_general
    special = ["plain"] * nkeys
    for j in range(nkeys):
        q = rng.random()
        if q < 0.15 and n > 1:
            data[f"k{j}"] = np.ma.array(data[f"k{j}"], mask=rng.random(n) < 0.15); special[j] = "masked"
        elif q < 0.25 and styles[j] in ("dense", "gappy", "single") and kinds[j] != "bool":
            special[j] = "categorical"
        elif q < 0.32 and like_gpu and nkeys == 1:
            f = rng.integers(-3, 4, n) * 0.25
            if rng.random() < 0.5: f[rng.random(n) < 0.1] = np.nan
            data[f"k{j}"] = f; special[j] = "float"; kinds[j] = "f8"
    v = rng.normal(0, 3, n)
    if rng.random() < 0.5: v[rng.random(n) < 0.2] = np.nan
    vdt = str(rng.choice(["i1", "i2", "i4", "i8", "u1", "u2", "u4"]))
    # (values that fit the type: a negative number wrapped into uint32 is ~4e9, its variance a difference of two ~2e19 moments — rounding noise on
    #  either side, not a result to compare)
    vi = rng.integers(0, 200, n) if vdt.startswith("u") else rng.integers(-100 if vdt == "i1" else -1000, 100 if vdt == "i1" else 1000, n)
    data.update(v=v, vi=vi.astype(vdt), vf=rng.normal(0, 1, n).astype("f4"))
    data.update(vm=np.ma.array(rng.normal(1, 2, n), mask=rng.random(n) < 0.2), vim=np.ma.array(rng.integers(-50, 50, n), mask=rng.random(n) < 0.2))   # (values with missing entries)
    df = vaex.from_arrays(**data)
    for j in range(nkeys):
        if special[j] == "categorical":
            if rng.random() < 0.5:
                df.categorize(f"k{j}", inplace=True)
            else:   # labels of its own, one more than the codes need (a category without a row)
                top = int(np.max(data[f"k{j}"])) if n else 0
                df.categorize(f"k{j}", labels=[f"L{q:03d}" for q in rng.permutation(top + 2)], min_value=0, inplace=True) if int(np.min(data[f"k{j}"])) == 0 else df.categorize(f"k{j}", inplace=True)
    df["virt"] = df.vf * 2 + 1      # (round 6: a virtual column as a value — materialised once by vaex's own evaluate)
    df["alias"] = df.v              # (... and an alias of a real column)
    aggs = {"c": A.count(), "cv": A.count("v"), "s": A.sum("v"), "m": A.mean("v"), "sd": A.std("v"), "va": A.var("vi"), "lo": A.min("v"), "hi": A.max("vi"),
            "si": A.sum("vi"), "mf": A.mean("vf"), "sf": A.sum("vf"), "lof": A.min("vf"), "cs": A.count(selection="v > 0"), "ms": A.mean("vi", selection="vf < 0"),
            # round 6: arithmetic over aggregators, virtual columns
            "r": A.sum("v") / A.count(), "dd": A.max("vi") - A.min("vi"), "ng": -A.mean("vf"), "x3": 3 * A.sum("vi"), "svt": A.sum("virt"), "mal": A.mean("alias"),
            # round 6 (late): values with missing entries, nunique (a second device groupby)
            "svm": A.sum("vm"), "mvm": A.mean("vm"), "cvm": A.count("vm"), "svim": A.sum("vim"), "mvim": A.mean("vim"), "sdvm": A.std("vm")}
    if like_gpu:
        aggs.update(nu=A.nunique("vi"), nuk=A.nunique("vim"))
    pick = [str(p) for p in rng.choice(list(aggs), size=int(rng.integers(1, 5)), replace=False)]
    agg = {p: aggs[p] for p in pick}
    kw = dict(sort=True, ascending=bool(rng.random() < 0.5)) if rng.random() < 0.4 else {}
    d = df
    if rng.random() < 0.25 and n > 3: d = df[1:n - 1]
    if rng.random() < 0.25: d = d[d.vf > -0.5]
    delayed = bool(rng.random() < 0.25)
    keys = list(data)[:nkeys]
    by = keys if nkeys > 1 else keys[0]
    make_by = lambda: by
    # round 6: the key as a binner OBJECT (vaex.groupby.Grouper / BinnerInteger): the order is the object's, not the call's
    use_object = nkeys == 1 and not delayed and rng.random() < 0.2 and special[0] in ("plain", "masked")
    if use_object:
        okw = dict(sort=bool(rng.random() < 0.7), ascending=bool(rng.random() < 0.5))
        cls = vaex.groupby.BinnerInteger if kinds[0] in ("i1", "u1", "bool") else vaex.groupby.Grouper
        make_by = lambda: cls(d[keys[0]], **okw)
        kw, kwc = {}, (okw if okw["sort"] or cls is vaex.groupby.BinnerInteger else {})
        if cls is vaex.groupby.BinnerInteger: kwc = dict(sort=True, ascending=not (okw["sort"] and not okw["ascending"]))
    else:
        kwc = kw
    what = (seed, n, kinds, styles, special, pick, kw, "filtered" if d.filtered else "", "delayed" if delayed else "", ("object", okw) if use_object else "")
    try:
        want = original(d, make_by(), agg=agg, **kw)
    except Exception as e:
        want = e
    try:
        vg.last.clear()
        if delayed:
            p = d.groupby(by, agg=agg, delay=True, **kw)
            d.execute()
            got = p.get()
        else:
            got = d.groupby(make_by(), agg=agg, **kw)
    except Exception as e:
        got = e
        if only:
            import traceback
            traceback.print_exc(file=sys.stdout)
    kw = kwc
    paths[vg.last.get("path")] = paths.get(vg.last.get("path"), 0) + 1
    if isinstance(want, Exception):
        if isinstance(got, Exception) and type(got) is type(want):
            known["raises alike"] = known.get("raises alike", 0) + 1
        elif isinstance(want, IndexError) and not isinstance(got, Exception):
            known["reference: IndexError (BinnerInteger, descending / unsigned)"] = known.get("reference: IndexError (BinnerInteger, descending / unsigned)", 0) + 1
        else:
            bad.append((what, "the reference raises", type(want).__name__, str(want)[:100], "here", type(got).__name__, str(got)[:200]))
        continue
    if isinstance(got, Exception):
        if nkeys >= 2 and "masked" in special and isinstance(got, IndexError) and vg.last.get("path") != "device":
            # a call the wrapper DECLINED: vaex's own combined groupers over a key with missing values (see below: not deterministic there) — one of their outcomes is
            # an IndexError in GrouperCombined's `parent.bin_values.take(indices)` (vaex/groupby.py:383; seed 24616, once in three runs of the same process)
            label = "reference: combined groupers over a key with missing values (not deterministic there)"
            known[label] = known.get(label, 0) + 1
            continue
        bad.append((what, "raises here only", type(got).__name__, str(got)[:300])); continue
    wraps = any(st == "extreme" and kd in ("i2", "i4", "u2", "u4") for st, kd in zip(styles, kinds))
    if (len(want) == 0 or (wraps and "masked" in special and len(want) < len(got))) and len(d) > 0 and len(got) > 0 and vg.last.get("path") == "device":   # (with a missing-value group the reference keeps that one)
        known["reference: no group at all (key range wraps in the key's dtype)"] = known.get("reference: no group at all (key range wraps in the key's dtype)", 0) + 1
        continue
    if nkeys >= 2 and "masked" in special:
        # several keys, one with missing values: where the reference COMBINES its groupers (rows / cells < 10) the missing rows of a key it simplified to
        # BinnerInteger(min_value != 0) get the code N - min_value (vaex/groupby.py:203: fillmissing(N), :525 subtracts min_value) — they fall into OTHER
        # cells, and its answer varies from run to run (1697 / 1699 groups over the same 5000 rows: tools/r07_masked_combined.py).  The device's groups are
        # checked against the rows themselves instead
        label = "reference: combined groupers over a key with missing values (not deterministic there)"
        if vg.last.get("path") == "device":
            from collections import Counter
            def as_handed_back(j, values):   # (a categorical key comes back as its LABELS: the rows hold the codes)
                if special[j] != "categorical":
                    return values
                labels, first = list(df.category_labels(keys[j])), df.category_offset(keys[j])
                return [None if q is None else labels[q - first] for q in values]
            truth = Counter(zip(*[as_handed_back(j, d[k].to_numpy().tolist()) for j, k in enumerate(keys)]))
            mine = list(zip(*[got[k].tolist() for k in keys]))
            if set(mine) != set(truth) or len(mine) != len(truth) or ("c" in pick and dict(zip(mine, got["c"].tolist())) != dict(truth)):
                bad.append((what, "device groups are not the rows' key combinations", len(mine), len(truth)))
        known[label] = known.get(label, 0) + 1
        continue
    if "bool" in kinds and kw.get("sort") and not kw.get("ascending") and vg.last.get("path") == "device":
        known["reference: bool key descending, labels not reversed"] = known.get("reference: bool key descending, labels not reversed", 0) + 1
        continue
    try:
        if kw.get("sort") and nkeys == 1:
            same({c: got[c].to_numpy() for c in got.get_column_names()}, {c: want[c].to_numpy() for c in want.get_column_names()})
        else:
            same(cols(got, keys), cols(want, keys))
    except AssertionError as e:
        bad.append((what, vg.last.get("path"), vg.last.get("kernel"), str(e)[:300]))
        if only:
            print("why", vg.last)
            for label, t in (("got", got), ("want", want)):
                print(label, {c: t.sort(keys)[c].tolist()[:40] for c in t.get_column_names()})
print("calls", ncalls, "| answered by:", paths, "| groupby stats:", {k: vg.stats[k] for k in ("device", "task", "vaex")})
print("recognised reference defects / alike exceptions:", known)
print("columns coded on the device:", vg.stats.get("coded_on_device", 0), "on the host for want of room / a plain dtype:", vg.stats.get("coded_on_host", 0))
for b in bad[:15]:
    print("BAD", b)
assert not bad, len(bad)
print("DONE")
'''


def _run(gpu, ncalls, timeout):
    env = dict(os.environ, VAEX_NUM_THREADS=os.environ.get("VAEX_NUM_THREADS", "4"))
    out = subprocess.run([sys.executable, "-c", SCRIPT % dict(pkg=PKG, fake=FAKE, root=ROOT, gpu=gpu, ncalls=ncalls)], cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0 and "DONE" in out.stdout, out.stdout[-6000:] + out.stderr[-3000:]
    report = os.environ.get("VAEX_AMD_REPORT_DIR")
    if report and gpu:
        with open(os.path.join(report, "random_groupby_report.txt"), "w") as f:
            f.write("\n".join(line for line in out.stdout.splitlines() if line.startswith(("calls", "recognised", "columns coded", "DONE"))))
    return out.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_random_groupbys_host_logic_against_the_original():
    _run(0, 250, 900)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_random_groupbys_agree_with_the_original():
    _run(1, int(os.environ.get("VAEX_AMD_RANDOM_GROUPBYS", "700")), 1500)   # (a soak run raises it)
