"""Drop-in check with the REAL vaex Python package: the reference's own pure-Python modules (oracle/_ref/vaexpy, built by
oracle/build_ref.sh from /root/reference; it travels to the GPU box like the reference-built .so files) on top of the
stand-in third-party modules of oracle/fake.  vaex_amd.install() plugs the HIP classes in; an unmodified
df.count / df.mean / df.std / df.groupby then builds OUR binners, Grid and aggregators through vaex's own decode path
(vaex/cpu.py:44-65, :630-667, vaex/agg.py:278-321 — incl. the exact-size memory check) and reaches Grid.bin.

  * without a GPU (here): the hot-path calls must fail loudly (no CPU fallback inside the library), while aggregations the
    HIP classes do not offer (string / object aggregators, ...) must keep working on vaex's own C++ (the per-task fallback of install());
  * with one (-m gpu): the results must equal the CPU reference computed in the same process after uninstall()."""
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
import sys, time, numpy as np
sys.path[:0] = [%(pkg)r, %(fake)r, %(root)r]
import vaex, vaex.hash, vaex_amd
cpu = vaex.superagg
backend = vaex_amd.install()
hip = vaex_amd.superagg
# (no knob is touched: the library's defaults are the reference's behaviour, incl. AggFirst / AggList's block-local keep-mask index,
# src/agg_first.cpp:131, and AggNUnique's `count -= null_count` ROWS, src/agg_nunique.cpp:31-34 — both exercised below)
# which backend every task really ran on: each decode of the registered "aggregations" task part is recorded
import vaex_amd.vaexfast as _vf
used = []
_task = vaex_amd._installed["task_hip"]
_decode = _task.decode.__func__
def _recording_decode(cls, *a, **k):
    part = _decode(cls, *a, **k)
    used.append(("agg", part.backend_used, [type(ag).__module__ for d in part.aggregations for ag in d[2]]))
    return part
_task.decode = classmethod(_recording_decode)
_legacy = vaex.vaexfast.statisticNd_f8
assert _legacy is _vf.statisticNd_f8
def _recording_legacy(*a, **k):
    used.append(("legacy", "hip", []))
    return _legacy(*a, **k)
vaex.vaexfast.statisticNd_f8 = _recording_legacy
assert vaex.superagg is backend and sys.modules["vaex.superagg"] is backend
assert vaex.superagg.Grid is hip.Grid and vaex.superagg.AggSum_float64 is hip.AggSum_float64
assert hasattr(vaex.superagg, "AggNUnique_float64") and hasattr(vaex.superagg, "BinnerHash_int64") and hasattr(vaex.superagg, "AggFirst_float64_int64")
assert not hasattr(vaex.superagg, "AggCount_string")
assert vaex.hash.ordered_set_int64.__module__ == "vaex_amd.hashset"
rng = np.random.default_rng(1)
n = %(n)d
x = rng.normal(0, 1, n); x[::1000] = np.nan
kf = rng.integers(0, 50, n).astype("f8"); kf[::777] = np.nan
im = np.ma.array(rng.integers(0, 500, n).astype("i4"), mask=rng.random(n) < 0.03)   # thousands of missing rows per group
df = vaex.from_arrays(x=x, y=rng.normal(0, 1, n), v=rng.normal(3, 2, n), k=rng.integers(0, 9, n), kb=rng.integers(-10**12, 10**12, n) // 10**9 * 10**9,
                      kf=kf, i=rng.integers(-100, 100, n).astype("i4"), im=im, f4=rng.choice(np.array([0.1, 0.3, 0.30000001, 0.5, 0.7], dtype="f4"), n), s=np.array(["a", "bb", None, "dddd"], dtype=object)[rng.integers(0, 4, n)])
lim2 = [[-4, 4], [-4, 4]]
def two_keys(d):
    k, i = d["k"].to_numpy(), d["i"].to_numpy()
    o = np.lexsort((i, k))
    return [np.asarray(d[c].to_numpy(), dtype="f8")[o] for c in ["k", "i", "c", "s"]]
def by_key(d, key, cols):
    d = d.sort(key)
    return [np.asarray(d[c].to_numpy(), dtype="f8") for c in [key] + cols]
hot = {
  "count": lambda d: d.count(binby=["x", "y"], limits=lim2, shape=32),
  "count_edges": lambda d: d.count(binby=["x", "y"], limits=lim2, shape=16, edges=True),
  "mean_sel": lambda d: d.mean("v", binby=["x"], limits=[-4, 4], shape=16, selection="v > 3"),
  "std": lambda d: d.std("v", binby=["x", "y"], limits=lim2, shape=8),
  "sum_i4": lambda d: d.sum("i", binby=["y"], limits=[-4, 4], shape=8),
  "minmax_binned": lambda d: np.stack([d.min("v", binby="x", limits=[-3, 3], shape=8), d.max("v", binby="x", limits=[-3, 3], shape=8)]),
  "limits_none": lambda d: d.count(binby="y", shape=8),            # legacy statisticNd pass for the limits (vaex/cpu.py:533-538)
  "minmax": lambda d: d.minmax("v"),
  "groupby_small": lambda d: by_key(d.groupby("k", agg={"s": vaex.agg.sum("v"), "c": vaex.agg.count(), "m": vaex.agg.mean("v"), "sd": vaex.agg.std("v")}), "k", ["s", "c", "m", "sd"]),
  "groupby_sparse": lambda d: by_key(d.groupby("kb", agg={"s": vaex.agg.sum("v"), "c": vaex.agg.count()}), "kb", ["s", "c"]),
  "groupby_float_nan": lambda d: by_key(d.groupby("kf", agg={"c": vaex.agg.count()}), "kf", ["c"]),
  "first_last": lambda d: np.stack([d.first("v", "y", binby="x", limits=[-4, 4], shape=8), d.last("v", "y", binby="x", limits=[-4, 4], shape=8)]),  # AggFirst_float64_float64
  "groupby_two_keys": lambda d: two_keys(d.groupby(["k", "i"], agg={"c": vaex.agg.count(), "s": vaex.agg.sum("v")})),  # GrouperCombined: vaex/groupby.py:526-584
}
hot["nunique"] = lambda d: d._compute_agg("nunique", "i", binby="y", limits=[-4, 4], shape=4)   # AggNUnique_int32
hot["groupby_nunique"] = lambda d: by_key(d.groupby("k", agg={"u": vaex.agg.nunique("i"), "uv": vaex.agg.nunique("kf", dropnan=True)}), "k", ["u", "uv"])
hot["groupby_list"] = lambda d: (lambda g: [sorted(c) for c in g["l"].tolist()])(d.groupby("k", agg={"l": vaex.agg.list("i")}).sort("k"))   # AggList_int32_int64
# the reference's quirks, reproduced by default: the selection reaches AggFirst as a keep-mask it reads block-locally
# (src/agg_first.cpp:131); dropmissing / dropnan take the number of missing / NaN ROWS away (src/agg_nunique.cpp:31-34)
hot["first_sel"] = lambda d: np.stack([np.ma.filled(d.first("v", "y", binby="x", limits=[-4, 4], shape=8, selection="v > 3"), -1e300),
                                        np.ma.filled(d.last("v", "y", binby="x", limits=[-4, 4], shape=8, selection="y < 0"), -1e300)])
hot["groupby_nunique_drop"] = lambda d: by_key(d.groupby("k", agg={"um": vaex.agg.nunique("im", dropmissing=True), "ua": vaex.agg.nunique("kf", dropna=True),
                                                                    "u0": vaex.agg.nunique("im")}), "k", ["um", "ua", "u0"])
# (dropmissing: the slots the reference appends per unselected row are uninitialised memory, src/agg_list.cpp:68-71)
hot["groupby_list_sel"] = lambda d: (lambda g: [sorted(c) for c in g["l"].tolist()])(d.groupby("k", agg={"l": vaex.agg.list("i", selection="v > 5", dropmissing=True)}).sort("k"))
# selections: comparison expressions run as device predicates (vaex_amd/vaex_selection.py); so does a named selection whose history
# resolves to one (tests/test_vaex_named_selection.py)
hot["count_sel2"] = lambda d: d.count(binby=["x", "y"], limits=lim2, shape=16, selection="(x > 0) & (v < 3.5)")
hot["sum_sel_int"] = lambda d: d.sum("v", binby="y", limits=[-4, 4], shape=8, selection="~(i >= 3) | (x < -1)")
hot["f32_boundary"] = lambda d: d.count(binby="y", limits=[-4, 4], shape=4, selection="f4 <= 0.3")   # numpy compares in float32: float32(0.3) <= 0.3
def _named(d):
    d.select("v > 4")
    return d.count(binby="x", limits=[-4, 4], shape=8, selection=True)
hot["named_sel"] = _named
def _mixed(d):   # ONE task, five aggregations: a device predicate, a named selection (resolved: device too), none, a list of selections (host masks)
    d.select("v > 4")
    a = d.count(binby="x", limits=[-4, 4], shape=8, selection="(v > 3) & (y < 1)", delay=True)
    b = d.sum("v", binby="x", limits=[-4, 4], shape=8, selection=True, delay=True)
    c = d.mean("v", binby="x", limits=[-4, 4], shape=8, delay=True)
    e = d.count(binby="x", limits=[-4, 4], shape=8, selection=["v > 3", "i < 0"], delay=True)
    f = d.max("i", binby="x", limits=[-4, 4], shape=8, selection="i < 50", delay=True)
    d.execute()
    return [np.asarray(a.get()), np.asarray(b.get()), np.asarray(c.get()), np.asarray(e.get()), np.asarray(f.get())]
hot["mixed_selections"] = _mixed
fallback = {   # not offered by the HIP classes: must run on vaex's own C++ after install(), GPU or not
  "count_string": lambda d: d.count("s", binby="y", limits=[-4, 4], shape=4),   # AggCount_string
  "count_string_sel": lambda d: d.count("s", binby="y", limits=[-4, 4], shape=4, selection="v > 3"),   # ... with a planned predicate: numpy inside process
}
has_gpu = hip.device_count() > 0
got = {}
if not has_gpu:
    for name, fn in hot.items():
        try:
            fn(df)
        except RuntimeError as e:
            assert "no HIP device" in str(e), (name, e)
            print("ok-loud-failure", name)
        else:
            raise SystemExit("computed without a GPU: " + name)
else:
    from vaex_amd import vaex_groupby as vg, vaex_selection as vsel
    seen_device = 0
    for name, fn in hot.items():
        del used[:]
        vg.last.clear()
        got[name] = fn(df)
        # every task of a hot call ran on the HIP classes (and there was at least one: an aggregation task part or the legacy statistic)
        # — or the call was a groupby answered by the device groupby as a whole (vaex_amd/vaex_groupby.py: no vaex task at all)
        whole = vg.last.get("path") == "device"
        assert (used or whole) and all(u[1] == "hip" for u in used), (name, used, vg.last)
        assert all(m == "vaex_amd.superagg" for u in used for m in u[2]), (name, used)
        assert whole == (name in ("groupby_small", "groupby_sparse", "groupby_two_keys", "groupby_float_nan")), (name, vg.last)   # (round 6, late: float keys are grouped by their bit patterns)
        if whole:
            hashed = name in ("groupby_sparse", "groupby_float_nan")
            assert ("gb_scatter" in vg.last["kernel"]) == hashed and ("part_scatter" in vg.last["kernel"] or "bin_" in vg.last["kernel"] or hashed), (name, vg.last)
        print("ok-backend hip", name, len(used), vg.last.get("kernel", ""))
        if name in ("mean_sel", "count_sel2", "sum_sel_int", "f32_boundary", "mixed_selections", "named_sel"):
            assert vsel.stats["device_chunks"] > seen_device, (name, vsel.stats)   # the predicate ran on the device
        elif name in ("first_sel", "groupby_list_sel"):   # (aggregators that read host masks only)
            assert vsel.stats["device_chunks"] == seen_device, (name, vsel.stats)  # vaex's own mask
        seen_device = vsel.stats["device_chunks"]
for name, fn in fallback.items():
    del used[:]
    got[name] = fn(df)
    assert used and all(u[1] == "cpu" for u in used) and all(m != "vaex_amd.superagg" for u in used for m in u[2]), (name, used)
    print("ok-backend cpu", name)
vaex.vaexfast.statisticNd_f8 = _legacy
vaex_amd.uninstall()
assert vaex.superagg is cpu and vaex.hash.ordered_set_int64.__module__ != "vaex_amd.hashset"
df2 = vaex.from_arrays(**{c: df[c].to_numpy() for c in df.get_column_names()})
CANCEL = {"std": None, "groupby_small": {4}}   # {call: elements of its result under the cancellation bound (None: all)} — variances / standard deviations only
def same(a, b, name, j=None):
    if isinstance(a, (list, tuple)):
        assert len(a) == len(b), name
        for i, (p, q) in enumerate(zip(a, b)):
            same(p, q, name, i if j is None else j)
        return
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.dtype.kind in "iub":   # counts, keys, integer sums, integer min / max: exact
        assert np.array_equal(a, b), name
    elif name in CANCEL and (CANCEL[name] is None or j in CANCEL[name]):   # std / var: a difference of two large moments (bound of tests/cases.py)
        assert np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True), (name, j, np.nanmax(np.abs(a - b)))
    else:   # fp64 sums / means / extrema: 1e-12 of the value or of the result's largest magnitude (<= sum|v| of that cell)
        fin = np.abs(b[np.isfinite(b)])
        scale = max(float(fin.max()) if fin.size else 0.0, 1.0)
        assert np.allclose(a, b, rtol=1e-12, atol=1e-12 * scale, equal_nan=True), (name, j, np.nanmax(np.abs(a - b)))
for name in list(got):
    fn = hot.get(name) or fallback[name]
    same(got[name], fn(df2), name)
    print("ok-parity" if name in hot else "ok-fallback", name)
if has_gpu and %(timing)d:
    # the host-streamed regime an actual vaex user hits: numpy columns, vaex's executor chunks them (1 Mi rows) over its
    # thread pool, every chunk crosses PCIe
    m = %(timing)d
    big = vaex.from_arrays(x=rng.normal(0, 1, m), y=rng.normal(0, 1, m))
    def run():
        t0 = time.perf_counter()
        c = big.count(binby=["x", "y"], limits=lim2, shape=256)
        return time.perf_counter() - t0, c
    cpu_t = min(run()[0] for _ in range(2))
    vaex_amd.install(chunk_size=None)   # vaex's own chunk bracket (1 Mi rows)
    run()
    hip_t, c = min((run() for _ in range(3)), key=lambda r: r[0])
    assert int(c.sum()) <= m
    print("TIMING vaex df.count(binby=[x,y], shape=256) on %%d host rows: hip %%.1f ms = %%.2f Grows/s (%%.1f GB/s over PCIe), cpu (reference C++, %%d threads) %%.1f ms = %%.2f Grows/s"
          %% (m, hip_t * 1e3, m / hip_t / 1e9, m * 16 / hip_t / 1e9, vaex.settings.main.thread_count, cpu_t * 1e3, m / cpu_t / 1e9))
    # the columns registered with the device column cache: the first pass crosses PCIe, later passes find their chunks in HBM
    assert vaex_amd.cache_columns(big) == 16 * m
    first = run()[0]
    again, c2 = min((run() for _ in range(3)), key=lambda r: r[0])
    assert np.array_equal(c, c2)
    stats = hip.cache_stats()
    assert stats["hits"] > 0 and stats["bytes"] == 16 * m, stats
    print("TIMING   same call, columns registered (vaex_amd.cache_columns): first pass %%.1f ms, later passes %%.1f ms = %%.2f Grows/s (chunks of %%d rows served from HBM)"
          %% (first * 1e3, again * 1e3, m / again / 1e9, vaex.settings.main.chunk.size_max))
    vaex_amd.uninstall()
    vaex_amd.install()   # the default: chunk_size="auto" raises vaex's upper chunk bracket to 64 Mi rows (one chunk per pool thread)
    assert vaex.settings.main.chunk.size_max == vaex_amd.AUTO_CHUNK_ROWS_MAX
    run()
    big_t, c3 = min((run() for _ in range(3)), key=lambda r: r[0])
    assert np.array_equal(c, c3)
    print("TIMING   with plain install() (chunk bracket raised to %%d rows): cached passes %%.1f ms = %%.2f Grows/s" %% (vaex.settings.main.chunk.size_max, big_t * 1e3, m / big_t / 1e9))
    vaex_amd.uncache_columns()
    vaex_amd.uninstall()
    assert vaex.settings.main.chunk.size_max == 1024 ** 2
    vaex_amd.install()
    host_t2, c4 = min((run() for _ in range(3)), key=lambda r: r[0])
    assert np.array_equal(c, c4)
    print("TIMING   host-streamed with plain install(): %%.1f ms = %%.2f Grows/s (%%.1f GB/s over PCIe)" %% (host_t2 * 1e3, m / host_t2 / 1e9, m * 16 / host_t2 / 1e9))
    vaex_amd.uncache_columns()
    vaex_amd.uninstall()
    # a FILTERED frame, big[big.x > 0]: vaex copies every column of every chunk through a boolean index before a task part sees a row
    # (vaex/execution.py:515-523) — fresh temporaries, so the device column cache never hits; with install() the chunks stay uncompacted
    # and the filter is a device predicate in the aggregators' keep-mask (vaex_amd/vaex_filter.py)
    flt = big[big.x > 0]
    def run_f():
        t0 = time.perf_counter()
        c = flt.count(binby=["x", "y"], limits=lim2, shape=256)
        return time.perf_counter() - t0, c
    cpu_f, cf = min((run_f() for _ in range(2)), key=lambda r: r[0])
    vaex_amd.install(filters=False)
    vaex_amd.cache_columns(big, ["x", "y"])
    run_f()
    old_f, c5 = min((run_f() for _ in range(2)), key=lambda r: r[0])
    vaex_amd.uncache_columns()
    vaex_amd.uninstall()
    vaex_amd.install()
    vaex_amd.cache_columns(big, ["x", "y"])
    run_f()
    new_f, c6 = min((run_f() for _ in range(3)), key=lambda r: r[0])
    assert np.array_equal(cf, c5) and np.array_equal(cf, c6) and int(cf.sum()) > m // 3
    print("TIMING vaex filtered frame big[big.x > 0].count(binby=[x,y], shape=256) on %%d host rows, columns registered: cpu (reference, %%d threads) %%.1f ms; install(filters=False) = vaex's numpy compaction per chunk %%.1f ms = %%.2f Grows/s; install() = filter as a device predicate over uncompacted chunks %%.1f ms = %%.2f Grows/s"
          %% (m, vaex.settings.main.thread_count, cpu_f * 1e3, old_f * 1e3, m / old_f / 1e9, new_f * 1e3, m / new_f / 1e9))
    vaex_amd.uncache_columns()
    vaex_amd.uninstall()
    # df.groupby(k).agg(sum / mean / std) of an unmodified vaex: its own two passes on the CPU (2e7-row slice), the device groupby
    # behind the same call on all rows — host columns (every call crosses PCIe), then registered columns (HBM-resident copies)
    from vaex_amd import vaex_groupby as vg
    big["k"] = rng.integers(0, 1_000_000, m)
    big["v"] = rng.normal(3, 2, m)
    spec = {"s": vaex.agg.sum("v"), "m": vaex.agg.mean("v"), "sd": vaex.agg.std("v")}
    small = big[:20_000_000].extract()
    t0 = time.perf_counter(); g_cpu = small.groupby("k", agg=spec); cpu_t = time.perf_counter() - t0
    vaex_amd.install()
    def run_g(d):
        t0 = time.perf_counter()
        g = d.groupby("k", agg=spec)
        return time.perf_counter() - t0, g
    run_g(big)
    host_t, g = min((run_g(big) for _ in range(2)), key=lambda r: r[0])
    assert vg.last["path"] == "device" and len(g) == 1_000_000
    g_small = small.groupby("k", agg=spec)
    a, b = g_small.sort("k"), g_cpu.sort("k")
    assert np.array_equal(np.ma.getdata(a["k"].to_numpy()), np.ma.getdata(b["k"].to_numpy())) and np.allclose(a["s"].to_numpy(), b["s"].to_numpy(), rtol=1e-12, atol=1e-9) and np.allclose(a["sd"].to_numpy(), b["sd"].to_numpy(), rtol=1e-9, atol=1e-9)
    vaex_amd.cache_columns(big, ["k", "v"])
    run_g(big)
    dev_t, g2 = min((run_g(big) for _ in range(3)), key=lambda r: r[0])
    print("TIMING vaex df.groupby(k, agg=sum/mean/std), 1e6 int64 keys: cpu (reference, %%d threads) %%.0f ms on %%d rows = %%.3f Grows/s; device groupby on %%d host rows %%.1f ms = %%.2f Grows/s (%%s); columns registered: %%.1f ms = %%.2f Grows/s"
          %% (vaex.settings.main.thread_count, cpu_t * 1e3, len(small), len(small) / cpu_t / 1e9, m, host_t * 1e3, m / host_t / 1e9, vg.last["kernel"], dev_t * 1e3, m / dev_t / 1e9))
    vaex_amd.uncache_columns()
    vaex_amd.uninstall()
'''


def _run(n, timing, timeout):
    env = dict(os.environ, VAEX_NUM_THREADS=os.environ.get("VAEX_NUM_THREADS", "4"))
    out = subprocess.run([sys.executable, "-c", SCRIPT % dict(pkg=PKG, fake=FAKE, root=ROOT, n=n, timing=timing)], cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    return out.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_unmodified_vaex_without_a_gpu_fails_loudly_and_falls_back():
    import vaex_amd
    if vaex_amd.superagg.device_count() > 0:
        pytest.skip("a GPU is visible: see the -m gpu test")
    out = _run(20000, 0, 300)
    assert out.count("ok-loud-failure") == 24 and out.count("ok-fallback") == 2, out


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.isdir(os.path.join(PKG, "vaex")), reason="real vaex package not built (oracle/build_ref.sh needs /root/reference)")
def test_unmodified_vaex_drives_the_hip_classes_on_the_gpu():
    out = _run(300_000, int(os.environ.get("VAEX_DROPIN_TIMING_ROWS", "400000000")), 900)
    assert out.count("ok-parity") == 24 and out.count("ok-fallback") == 2, out
    assert out.count("ok-backend hip") == 24 and out.count("ok-backend cpu") == 2, out
    line = [l for l in out.splitlines() if l.startswith("TIMING")]
    assert line, out
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "vaex_dropin_timing.txt"), "w") as f:
        f.write("\n".join(line) + "\n")
    print("\n".join(line))
