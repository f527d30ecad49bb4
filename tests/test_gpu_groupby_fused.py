"""-m gpu: the fused hash groupby (vxh_groupby_run: radix partition + LDS-probed aggregation) against numpy in double on
the same rows — per key: rows, count / sum / sum of squares of the non-NaN values, mean, var, std — bit-exact for keys and
counts, 1e-12 of sum|v| (sum|v^2|) per group for the float sums; plus the merge of partial results, the edge cases (few
groups, one row, INT64_MIN as a key, every integer key dtype, two value columns) and the fallbacks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_DT = {"float64": 0, "float32": 1, "int64": 2, "int32": 3, "int16": 4, "int8": 5, "uint64": 6, "uint32": 7, "uint16": 8, "uint8": 9, "bool": 10}


def _want(keys, cols):
    uniq, inv = np.unique(keys.astype(np.int64) if keys.dtype != np.uint64 else keys.view(np.int64), return_inverse=True)
    out = dict(k=uniq, rows=np.bincount(inv, minlength=len(uniq)), v=[])
    for v in cols:
        ok = v == v
        out["v"].append(dict(cnt=np.bincount(inv[ok], minlength=len(uniq)), s=np.bincount(inv[ok], weights=v[ok], minlength=len(uniq)),
                             s2=np.bincount(inv[ok], weights=v[ok] * v[ok], minlength=len(uniq)), sabs=np.bincount(inv[ok], weights=np.abs(v[ok]), minlength=len(uniq))))
    return out


def _check(sa, res, want):
    assert len(res) == len(want["k"])
    np.testing.assert_array_equal(np.asarray(res.column(sa.GB_KEYS)), want["k"])
    np.testing.assert_array_equal(np.asarray(res.column(sa.GB_ROWS)), want["rows"])
    for j, w in enumerate(want["v"]):
        cnt, s, s2 = (np.asarray(res.column(c, j)) for c in (sa.GB_COUNT, sa.GB_SUM, sa.GB_SUM2))
        np.testing.assert_array_equal(cnt, w["cnt"])
        assert np.all(np.abs(s - w["s"]) <= 1e-12 * w["sabs"])
        assert np.all(np.abs(s2 - w["s2"]) <= 1e-12 * w["s2"])
        with np.errstate(divide="ignore", invalid="ignore"):
            mean = s / cnt
            var = s2 / cnt - mean ** 2
        # the derived columns are the SAME IEEE operations on the device's own sums: bit-exact against numpy on them
        np.testing.assert_array_equal(np.asarray(res.column(sa.GB_MEAN, j)), mean)
        np.testing.assert_array_equal(np.asarray(res.column(sa.GB_VAR, j)), var)
        np.testing.assert_array_equal(np.asarray(res.column(sa.GB_STD, j)), var ** 0.5)


@pytest.mark.parametrize("n,groups", [(1, 1), (1000, 7), (300_000, 5_000), (3_000_000, 1_000_000), (2_000_000, 1_900_000)])
def test_groupby_run_host_rows(sa, gpu_ready, n, groups):
    rng = np.random.default_rng(n + groups)
    k = (rng.integers(0, groups, n) * 2654435761) % (1 << 40) - (1 << 39)
    v = rng.normal(3, 2, n)
    v[rng.random(n) < 0.01] = np.nan
    res = sa.groupby_run(k, [v], _DT["int64"])
    _check(sa, res, _want(k, [v]))
    info = res.info()
    assert info["buckets"] >= 64 and info["slots"] >= 2048 and info["slots"] % 4 == 0  # (4-key lines, as many as the LDS holds)


def test_groupby_run_two_keys_only(sa, gpu_ready):
    """two distinct keys: every row lands in one of two buckets — the queue slack grows on retry (small inputs), and the
    result is still exact"""
    rng = np.random.default_rng(2)
    n = 200_000
    k = np.where(rng.random(n) < 0.3, 7, -(1 << 50)).astype(np.int64)
    v = rng.normal(0, 1, n)
    res = sa.groupby_run(k, [v], _DT["int64"])
    _check(sa, res, _want(k, [v]))
    assert res.info()["retries"] >= 1


def test_groupby_run_device_rows_two_values(sa, gpu_ready):
    import torch
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 20_000_000
    k = torch.randint(0, 400_000, (n,), dtype=torch.int64, device="cuda", generator=g) * 7919 - 10**9
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    w = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    w[::1000] = float("nan")
    res = sa.groupby_run(k, [v, w], _DT["int64"])  # (no synchronize: the library orders itself after the default stream)
    _check(sa, res, _want(k.cpu().numpy(), [v.cpu().numpy(), w.cpu().numpy()]))
    assert 2048 <= res.info()["slots"] < 4096  # (two value columns: 40-byte slots)


@pytest.mark.parametrize("dtype", ["int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"])
def test_groupby_run_key_dtypes(sa, gpu_ready, dtype):
    rng = np.random.default_rng(5)
    n = 100_000
    if dtype == "bool":
        k = rng.random(n) < 0.3
    else:
        info = np.iinfo(dtype)
        k = rng.integers(max(info.min, -3000), min(info.max, 3000), n).astype(dtype)
    v = rng.normal(0, 1, n)
    res = sa.groupby_run(k, [v], _DT[dtype])
    kk = k.astype(np.int64) if dtype != "uint64" else k.astype(np.int64)
    _check(sa, res, _want(kk, [v]))


def test_groupby_run_int64_min_key_and_empty(sa, gpu_ready):
    lo = np.iinfo(np.int64).min
    k = np.array([5, lo, 7, lo, 5, np.iinfo(np.int64).max, lo], dtype=np.int64)
    v = np.array([1.0, 2.0, np.nan, 4.0, 5.0, 6.0, np.nan])
    res = sa.groupby_run(k, [v], _DT["int64"])
    _check(sa, res, _want(k, [v]))
    assert np.asarray(res.column(sa.GB_KEYS))[0] == lo
    empty = sa.groupby_run(np.array([], dtype=np.int64), [np.array([], dtype=np.float64)], _DT["int64"])
    assert len(empty) == 0 and len(np.asarray(empty.column(sa.GB_KEYS))) == 0
    with pytest.raises(RuntimeError, match="float64 value columns"):
        sa.groupby_run(k, [v.astype("f4")], _DT["int64"])
    with pytest.raises(RuntimeError, match="integer key"):
        sa.groupby_run(v, [v], _DT["float64"])


def test_groupby_merge_partials(sa, gpu_ready):
    rng = np.random.default_rng(9)
    n = 1_500_000
    k = rng.integers(-200_000, 200_000, n) * 3
    v = rng.normal(1, 3, n)
    parts = [sa.groupby_run(k[i::3].copy(), [v[i::3].copy()], _DT["int64"]) for i in range(3)]
    cat = lambda c, j=0: np.concatenate([np.asarray(p.column(c, j)) for p in parts])
    merged = sa.groupby_merge(cat(sa.GB_KEYS), cat(sa.GB_ROWS), [cat(sa.GB_COUNT)], [cat(sa.GB_SUM)], [cat(sa.GB_SUM2)])
    _check(sa, merged, _want(k, [v]))


def test_frame_groupby_takes_the_fused_path_and_falls_back(sa, gpu_ready):
    import torch
    from vaex_amd.binned import Frame, agg
    g = torch.Generator(device="cuda").manual_seed(4)
    n = 6_000_000
    k = (torch.randint(0, 300_000, (n,), dtype=torch.int64, device="cuda", generator=g) * 2654435761) % (1 << 40)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    v[::999] = float("nan")
    spec = {"n": agg.count(), "c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
    df = Frame(dict(k=k, v=v))
    got = df.groupby("k", spec)
    assert df.last_groupby_info["buckets"] >= 64  # the fused path ran
    w = _want(k.cpu().numpy(), [v.cpu().numpy()])
    np.testing.assert_array_equal(got["k"], w["k"])
    np.testing.assert_array_equal(got["n"], w["rows"])
    np.testing.assert_array_equal(got["c"], w["v"][0]["cnt"])
    assert np.all(np.abs(got["s"] - w["v"][0]["s"]) <= 1e-12 * w["v"][0]["sabs"])
    # a heavy hitter (half of the rows on one key): too skewed for the partitioned pass by itself — the key is found in a sample and
    # peeled off (Frame._groupby_peeled), the rest takes the fused pass
    k2 = k.clone()
    k2[::2] = 12345678901
    df2 = Frame(dict(k=k2, v=v))
    df2.last_groupby_info = None
    got2 = df2.groupby("k", spec)
    assert df2.last_groupby_info is not None and df2.last_groupby_info.get("heavy_keys") == 1 and df2.last_groupby_info["retries"] == 0, df2.last_groupby_info
    w2 = _want(k2.cpu().numpy(), [v.cpu().numpy()])
    np.testing.assert_array_equal(got2["k"], w2["k"])
    np.testing.assert_array_equal(got2["n"], w2["rows"])
    np.testing.assert_array_equal(got2["c"], w2["v"][0]["cnt"])
    assert np.all(np.abs(got2["s"] - w2["v"][0]["s"]) <= 1e-12 * w2["v"][0]["sabs"])
    # ... without the sample (host rows are not sampled; here: switched off) the skewed call still answers: ordered_set + BinnerHash
    df3 = Frame(dict(k=k2, v=v))
    df3.heavy_key_rows = 1 << 62
    df3.last_groupby_info = None
    got3 = df3.groupby("k", spec)
    np.testing.assert_array_equal(got3["k"], w2["k"])
    np.testing.assert_array_equal(got3["c"], w2["v"][0]["cnt"])
    assert np.all(np.abs(got3["s"] - w2["v"][0]["s"]) <= 1e-12 * w2["v"][0]["sabs"])
    # min / max are outside the fused signature
    got3 = df.groupby("k", {"lo": agg.min("v"), "s": agg.sum("v")})
    np.testing.assert_array_equal(got3["k"], w["k"])


@pytest.mark.parametrize("groups", [3_400_000, 4_700_000])
def test_frame_groupby_millions_of_scattered_keys(sa, gpu_ready, groups):
    """more distinct keys than the default plan's 512 LDS tables hold: 3.4e6 -> the pass retries with 1024 buckets (whose
    bookkeeping leaves room for 4096-row tiles only); 4.7e6 -> beyond 1024 tables of ~4000 keys: the pass reports it and
    Frame.groupby answers through ordered_set + BinnerHash.  Either way the groups equal numpy's."""
    import torch
    from vaex_amd.binned import Frame, agg
    g = torch.Generator(device="cuda").manual_seed(groups)
    n = 3 * groups
    k = (torch.randint(0, groups, (n,), dtype=torch.int64, device="cuda", generator=g) * 2654435761) % (1 << 42)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    df = Frame(dict(k=k, v=v))
    df.last_groupby_info = None
    got = df.groupby("k", {"c": agg.count("v"), "s": agg.sum("v")})
    w = _want(k.cpu().numpy(), [v.cpu().numpy()])
    assert len(w["k"]) > 0.9 * groups
    np.testing.assert_array_equal(got["k"], w["k"])
    np.testing.assert_array_equal(got["c"], w["v"][0]["cnt"])
    assert np.all(np.abs(got["s"] - w["v"][0]["s"]) <= 1e-12 * w["v"][0]["sabs"])
    if groups < 4_000_000:
        assert df.last_groupby_info is not None and df.last_groupby_info["buckets"] == 1024 and df.last_groupby_info["retries"] >= 1
    else:
        assert df.last_groupby_info is None  # (the partitioned pass declined)


@pytest.mark.parametrize("where", ["host", "device"])
def test_groupby_run_with_a_keep_mask(sa, gpu_ready, where):
    """vxh_groupby_run_kept: rows outside the mask leave no record (a filtered frame / a selection over the whole call) — the result
    is that of the kept rows alone, groups without a kept row do not exist; mask values other than 1 do not keep"""
    import torch
    rng = np.random.default_rng(8)
    n = 2_000_000
    k = (rng.integers(0, 300_000, n) * 2654435761) % (1 << 40) - (1 << 39)
    v = rng.normal(3, 2, n); v[rng.random(n) < 0.01] = np.nan
    w = rng.normal(0, 1, n)
    keep = (rng.random(n) < 0.3).astype(np.uint8)
    keep[::1000] = 2   # (not 1: dropped, like Aggregator masks, src/agg_count.cpp:50)
    kept = keep == 1
    args = (k, [v, w], keep) if where == "host" else (torch.from_numpy(k).cuda(), [torch.from_numpy(v).cuda(), torch.from_numpy(w).cuda()], torch.from_numpy(keep).cuda())
    res = sa.groupby_run(args[0], args[1], _DT["int64"], keep=args[2])
    want = _want(k[kept], [v[kept], w[kept]])
    assert len(want["k"]) < len(np.unique(k))   # (some groups have no kept row)
    _check(sa, res, want)
    none = sa.groupby_run(args[0], args[1], _DT["int64"], keep=(np.zeros(n, dtype=np.uint8) if where == "host" else torch.zeros(n, dtype=torch.uint8, device="cuda")))
    assert len(none) == 0


def test_frame_groupby_with_a_selection_takes_the_fused_path(sa, gpu_ready):
    """Frame.groupby(selection=): one selection shared by the whole call rides the fused pass as its keep-mask (device and host rows)"""
    import torch
    from vaex_amd import binned
    rng = np.random.default_rng(9)
    n = 1_500_000
    k = (rng.integers(0, 200_000, n) * 2654435761) % (1 << 40)
    v = rng.normal(3, 2, n); v[::313] = np.nan
    x = rng.normal(0, 1, n)
    for device in (False, True):
        cols = dict(k=k, v=v, x=x)
        if device:
            cols = {c: torch.from_numpy(a).cuda() for c, a in cols.items()}
        f = binned.Frame(cols, superagg=sa)
        f.last_groupby_info = None
        got = f.groupby("k", {"c": binned.agg.count(), "cv": binned.agg.count("v"), "s": binned.agg.sum("v"), "m": binned.agg.mean("v"), "sd": binned.agg.std("v")}, selection="(x > 0.25) & (x < 2)")
        assert f.last_groupby_info is not None, "the fused pass did not run"
        kept = (x > 0.25) & (x < 2)
        want = _want(k[kept], [v[kept]])
        np.testing.assert_array_equal(got["k"], want["k"]); np.testing.assert_array_equal(got["c"], want["rows"]); np.testing.assert_array_equal(got["cv"], want["v"][0]["cnt"])
        assert np.all(np.abs(got["s"] - want["v"][0]["s"]) <= 1e-12 * want["v"][0]["sabs"])
        with np.errstate(divide="ignore", invalid="ignore"):
            mean = want["v"][0]["s"] / want["v"][0]["cnt"]
        assert np.allclose(got["m"], mean, rtol=1e-11, atol=0, equal_nan=True)


@pytest.mark.parametrize("keys", ["scattered", "dense_skewed"])
def test_aggregations_with_their_own_selection_keep_the_groups_of_all_rows(sa, gpu_ready, keys):
    """vaex.agg.count(selection=...) is NOT a filter — the groups are those of ALL rows, a group without a selected
    row reports count 0 / sum 0 / mean NaN (vaex/groupby.py:884-899).  The fused pass used to take such a selection as its keep-mask
    and the groups without a kept row vanished.  Scattered keys (the fused hash pass, run twice and joined) and a skewed dense range
    wider than one workgroup's LDS (the fused pass with the peel inside / the dense peel): half the groups have NO selected row."""
    import torch
    from vaex_amd import binned
    rng = np.random.default_rng(21)
    n = 5_000_000 if keys == "dense_skewed" else 1_200_000
    if keys == "scattered":
        base = rng.integers(0, 60_000, n)
        k = (base * 2654435761) % (1 << 40)
    else:
        base = np.where(rng.random(n) < 0.5, rng.integers(0, 8, n) * 3001, rng.integers(0, 40_000, n))   # eight heavy keys in a 40 000-cell range
        k = base + 17
    v = rng.normal(3, 2, n); v[::211] = np.nan
    x = rng.normal(0, 1, n)
    x[base % 2 == 1] = -1.0              # odd groups: no row passes "x > 0"
    for device in (True, False):
        cols = dict(k=k, v=v, x=x)
        if device:
            cols = {c: torch.from_numpy(a).cuda() for c, a in cols.items()}
        f = binned.Frame(cols, superagg=sa)
        A = binned.agg
        got = f.groupby("k", {"c": A.count(selection="x > 0"), "cv": A.count("v", selection="x > 0"), "s": A.sum("v", selection="x > 0"),
                              "m": A.mean("v", selection="x > 0"), "sd": A.std("v", selection="x > 0")})
        uniq, inv = np.unique(k, return_inverse=True)
        kept = x > 0
        ok = kept & (v == v)
        rows = np.bincount(inv[kept], minlength=len(uniq))
        cnt = np.bincount(inv[ok], minlength=len(uniq))
        s = np.bincount(inv[ok], weights=v[ok], minlength=len(uniq))
        sabs = np.bincount(inv[ok], weights=np.abs(v[ok]), minlength=len(uniq))
        np.testing.assert_array_equal(got["k"], uniq)                       # every group of ALL rows is there
        assert (rows == 0).sum() > len(uniq) // 3                          # ... and many of them have no selected row
        np.testing.assert_array_equal(got["c"], rows)
        np.testing.assert_array_equal(got["cv"], cnt)
        assert np.all(np.abs(got["s"] - s) <= 1e-12 * sabs)
        assert np.all(got["s"][cnt == 0] == 0)
        with np.errstate(divide="ignore", invalid="ignore"):
            mean = s / cnt
        assert np.allclose(got["m"], mean, rtol=1e-11, atol=0, equal_nan=True)
        assert np.isnan(got["m"][cnt == 0]).all() and np.isnan(got["sd"][cnt == 0]).all()
        # the same selection as the CALL's filter is the other thing: groups without a kept row do not exist
        flt = f.groupby("k", {"c": A.count()}, selection="x > 0")
        np.testing.assert_array_equal(flt["k"], uniq[rows > 0])
        np.testing.assert_array_equal(flt["c"], rows[rows > 0])
        if keys == "scattered":
            assert (f.last_groupby_info or {}).get("groups") is not None or f.last_groupby_info is not None


@pytest.mark.parametrize("flavour", ["zipf", "three_keys", "zipf_int32_selection"])
def test_frame_groupby_peels_heavy_keys(sa, gpu_ready, flavour):
    """skewed key columns (the head of a Zipf law, a handful of scattered keys, a default value) on the device: the heavy keys are found
    in a sample and aggregated as a dense groupby over their ordinals, the rest takes the fused pass with those rows masked out — same
    groups, counts bit-exact, sums to 1e-12 of sum|v|; with a selection whose mask holds values other than 1 (they do not keep)"""
    import torch
    from vaex_amd.binned import Frame, agg
    rng = np.random.default_rng(21)
    n = 5_000_000
    if flavour == "three_keys":
        k = rng.choice(np.array([-(1 << 45), 17, (1 << 50) + 3]), n, p=[0.6, 0.3, 0.1])
    else:
        z = rng.zipf(1.3, n)
        k = (np.minimum(z, 400_000) * 2654435761) % (1 << 40)
        if flavour == "zipf_int32_selection":
            k = (k % (1 << 31)).astype(np.int32)
    v = rng.normal(3, 2, n); v[::777] = np.nan
    keep = None
    if flavour == "zipf_int32_selection":
        keep = (rng.random(n) < 0.6).astype(np.uint8)
        keep[::500] = 3   # (not 1: the row is dropped)
    cols = dict(k=torch.from_numpy(np.ascontiguousarray(k)).cuda(), v=torch.from_numpy(v).cuda())
    if keep is not None:
        cols["sel"] = torch.from_numpy(keep).cuda()
    df = Frame(cols, superagg=sa)
    df.last_groupby_info = None
    spec = {"n": agg.count(), "c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
    got = df.groupby("k", spec, selection="sel" if keep is not None else None)
    info = df.last_groupby_info
    assert info is not None and info.get("heavy_keys", 0) >= (3 if flavour == "three_keys" else 2) and info["retries"] == 0, info
    kept = np.ones(n, dtype=bool) if keep is None else keep == 1
    w = _want(k[kept], [v[kept]])
    np.testing.assert_array_equal(np.asarray(got["k"]).astype(np.int64), w["k"])
    np.testing.assert_array_equal(got["n"], w["rows"])
    np.testing.assert_array_equal(got["c"], w["v"][0]["cnt"])
    assert np.all(np.abs(got["s"] - w["v"][0]["s"]) <= 1e-12 * w["v"][0]["sabs"])
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = w["v"][0]["s"] / w["v"][0]["cnt"]
        var = w["v"][0]["s2"] / w["v"][0]["cnt"] - mean ** 2
    assert np.allclose(got["m"], mean, rtol=1e-11, atol=0, equal_nan=True)
    big = w["v"][0]["cnt"] > 100
    assert np.allclose(np.asarray(got["sd"])[big] ** 2, var[big], rtol=1e-8, atol=0)
    # the same call again: the heavy keys are remembered per column object
    again = df.groupby("k", spec, selection="sel" if keep is not None else None)
    np.testing.assert_array_equal(again["n"], got["n"])


@pytest.mark.parametrize("where", ["device", "host"])
def test_frame_groupby_peels_heavy_keys_of_a_dense_range(sa, gpu_ready, where):
    """a dense key range too wide for one workgroup's LDS (the slab-partitioned pass) with a Zipf head and a default value: the heavy keys
    are peeled off here too (Frame._groupby_dense_peeled) — min / max and an int32 value column included, with a predicate selection"""
    import torch
    from vaex_amd.binned import Frame, agg
    rng = np.random.default_rng(22)
    n = 4_500_000
    k = np.minimum(rng.zipf(1.25, n), 300_000).astype(np.int64) - 1000
    v = rng.normal(3, 2, n); v[::555] = np.nan
    i = rng.integers(-1000, 1000, n).astype(np.int32)
    x = rng.normal(0, 1, n)
    cols = dict(k=k, v=v, i=i, x=x)
    if where == "device":
        cols = {c: torch.from_numpy(a).cuda() for c, a in cols.items()}
    spec = {"n": agg.count(), "c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v"), "lo": agg.min("v"), "hi": agg.max("i"), "si": agg.sum("i")}
    for selection in (None, "x > -0.5"):
        df = Frame(cols, superagg=sa)
        df.last_groupby_info = None
        got = df.groupby("k", spec, selection=selection)
        assert df.last_groupby_info is not None and df.last_groupby_info.get("dense") == 1 and df.last_groupby_info["heavy_keys"] >= 2, df.last_groupby_info
        plain = Frame(cols, superagg=sa)
        plain.heavy_key_rows = 1 << 62
        want = plain.groupby("k", spec, selection=selection)          # the same call without the peel
        kept = np.ones(n, dtype=bool) if selection is None else x > -0.5
        w = _want(k[kept], [v[kept]])
        np.testing.assert_array_equal(got["k"], w["k"]); np.testing.assert_array_equal(got["n"], w["rows"]); np.testing.assert_array_equal(got["c"], w["v"][0]["cnt"])
        assert np.all(np.abs(got["s"] - w["v"][0]["s"]) <= 1e-12 * w["v"][0]["sabs"])
        for name in ("k", "n", "c", "lo", "hi", "si"):
            np.testing.assert_array_equal(np.asarray(got[name]), np.asarray(want[name]), err_msg=name)
            assert np.asarray(got[name]).dtype == np.asarray(want[name]).dtype, name
        assert np.allclose(got["m"], want["m"], rtol=1e-11, atol=0, equal_nan=True) and np.allclose(got["sd"], want["sd"], rtol=1e-7, atol=1e-9, equal_nan=True)


def test_skewed_dense_key_range_takes_the_fused_pass_with_the_peel_inside(sa, gpu_ready):
    """round 4: heavy keys are peeled INSIDE the fused pass (vxh_groupby_run_peeled: gb_scatter looks every key up in an LDS copy of the
    heavy list, adds such rows to per-workgroup partials and leaves no record).  A dense key range wider than one workgroup's LDS whose
    sample shows heavy keys takes that pass too when the call is count / sum / mean / var / std of float64 columns; heavy keys listed
    but absent, a listed key with rows only outside the keep-mask, and NaN values of heavy rows are covered at the C-ABI level"""
    import torch
    from vaex_amd.binned import Frame, agg
    rng = np.random.default_rng(23)
    n = 4_500_000
    k = np.minimum(rng.zipf(1.25, n), 300_000).astype(np.int64) - 1000
    v = rng.normal(3, 2, n); v[::555] = np.nan
    cols = {c: torch.from_numpy(a).cuda() for c, a in dict(k=k, v=v).items()}
    spec = {"n": agg.count(), "c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
    df = Frame(cols, superagg=sa)
    df.last_groupby_info = None
    got = df.groupby("k", spec)
    info = df.last_groupby_info
    assert info is not None and info.get("dense_range_through_fused_pass") == 1 and info["heavy_keys"] >= 2 and info["retries"] == 0, info
    w = _want(k, [v])
    np.testing.assert_array_equal(got["k"], w["k"]); np.testing.assert_array_equal(got["n"], w["rows"]); np.testing.assert_array_equal(got["c"], w["v"][0]["cnt"])
    assert np.all(np.abs(got["s"] - w["v"][0]["s"]) <= 1e-12 * w["v"][0]["sabs"])
    # the C-ABI entry itself: two value columns, a keep-mask, a listed key that does not occur, one whose rows are all masked out
    m = 3_000_000
    kk = (rng.integers(0, 50_000, m).astype(np.int64) * 2654435761) % (1 << 40)
    hot, cold, masked_out = int(kk[0]), 12345678901234, int(kk[1])
    kk[rng.random(m) < 0.3] = hot
    v0 = rng.normal(1, 1, m); v1 = rng.normal(-2, 3, m); v0[::97] = np.nan
    keep = (rng.random(m) < 0.7).astype(np.uint8); keep[kk == masked_out] = 0
    res = sa.groupby_run(torch.from_numpy(kk).cuda(), [torch.from_numpy(v0).cuda(), torch.from_numpy(v1).cuda()], 2,   # (2 = VXH_I64)
                         keep=torch.from_numpy(keep).cuda(), heavy=np.array([hot, cold, masked_out, hot], dtype=np.int64))
    assert res.info()["heavy_keys_in_pass"] == 3 and res.info()["retries"] == 0, res.info()
    sel = keep == 1
    w2 = _want(kk[sel], [v0[sel], v1[sel]])
    _check(sa, res, w2)


def test_count_only_groupby_and_value_counts_take_the_fused_pass(sa, gpu_ready):
    """`n: count` over scattered int64 keys (and value_counts of a float column: its bit patterns are such keys) has no value column
    to carry through the partitioned pass: the key column lends its own 8 bytes as payload, only the row counts are read"""
    import torch
    from vaex_amd.binned import Frame, agg
    rng = np.random.default_rng(31)
    n = 3_000_000
    k = (rng.integers(0, 400_000, n) * 2654435761) % (1 << 42) - (1 << 41)
    x = rng.normal(0, 1, n).round(2); x[::501] = np.nan; x[::777] = -0.0
    for device in (True, False):
        cols = dict(k=k, x=x)
        if device:
            cols = {c: torch.from_numpy(a).cuda() for c, a in cols.items()}
        df = Frame(cols, superagg=sa)
        df.last_groupby_info = None
        got = df.groupby("k", {"n": agg.count()})
        assert df.last_groupby_info is not None, "the fused pass did not run"
        uniq, cnt = np.unique(k, return_counts=True)
        np.testing.assert_array_equal(got["k"], uniq); np.testing.assert_array_equal(got["n"], cnt)
        sel = df.groupby("k", {"n": agg.count()}, selection="x > 0.5")
        kept = x > 0.5
        uniq2, cnt2 = np.unique(k[kept], return_counts=True)
        np.testing.assert_array_equal(sel["k"], uniq2); np.testing.assert_array_equal(sel["n"], cnt2)
        vals, counts = df.value_counts("x")
        ok = x == x
        bits, c3 = np.unique(x[ok].view(np.int64), return_counts=True)
        want = dict(zip(bits.tolist(), c3.tolist()))
        got_pairs = {int(np.float64(v).view(np.int64)): int(c) for v, c in zip(vals, counts) if v == v}
        assert got_pairs == want
        assert int(counts[[i for i, v in enumerate(vals) if v != v][0]]) == int((~ok).sum())
        assert np.all(np.diff(counts) <= 0)


@pytest.mark.parametrize("case", ["bench_2e40", "negative_min", "exactly_32_bits", "too_wide", "int32_keys", "with_keep", "few_groups"])
def test_groupby_run_compact_records(sa, gpu_ready, case):
    """round 4: with the key RANGE known (vxh_groupby_run_ranged) the pass moves 12-byte records {remainder, value}: key - min is mixed
    by an invertible map whose top bits are the bucket, only the <= 32 bits below travel; gb_reduce mixes every group's key back.
    Same groups, counts bit-exact and sums within 1e-12 x sum|v| as with 16-byte records and as numpy — for ranges that fit, one that
    leaves exactly 32 bits, a negative minimum, narrow key dtypes, a keep-mask; a range too wide keeps the 16-byte records."""
    rng = np.random.default_rng(17)
    n = 3_000_000
    groups = 200_000
    base = rng.integers(0, groups, n)
    dtype = np.int64
    if case == "bench_2e40":
        keys = (base * 2654435761) % (1 << 40)
    elif case == "negative_min":
        keys = (base * 2654435761) % (1 << 38) - (1 << 37) - 12345
    elif case == "exactly_32_bits":   # 2^20 expected groups at 50 % load: 512 buckets = 9 bits; a 41-bit range leaves 32
        keys = (base * 2654435761) % (1 << 41)
        keys[0], keys[1] = 0, (1 << 41) - 1
    elif case == "too_wide":
        keys = ((base.astype(np.uint64) * np.uint64(11400714819323198485)) % np.uint64(1 << 62)).astype(np.int64)
    elif case == "int32_keys":
        keys = ((base * 2654435761) % (1 << 31) - (1 << 30)).astype(np.int32)
        dtype = np.int32
    elif case == "with_keep":
        keys = (base * 2654435761) % (1 << 40)
    else:
        keys = rng.choice(np.array([5, 1 << 35, -(1 << 33), 77], dtype=np.int64), n)
    keys = np.ascontiguousarray(keys.astype(dtype))
    v = rng.normal(3, 2, n)
    v[::97] = np.nan
    keep = (rng.random(n) < 0.5).astype(np.uint8) if case == "with_keep" else None
    kr = (int(keys.min()), int(keys.max()))
    dt = _DT[np.dtype(dtype).name]
    res = sa.groupby_run(keys, [v], dt, keep=keep, key_range=kr)
    assert res.info()["compact_records"] == (0 if case == "too_wide" else 1), (case, res.info())
    sa.config_set("gb_compact", 0)
    try:
        plain = sa.groupby_run(keys, [v], dt, keep=keep, key_range=kr)
        assert plain.info()["compact_records"] == 0
    finally:
        sa.config_set("gb_compact", 1)
    sel = slice(None) if keep is None else keep == 1
    want = _want(keys[sel], [v[sel]])
    _check(sa, res, want)
    _check(sa, plain, want)
    # on device-resident rows too
    import torch
    dev = sa.groupby_run(torch.from_numpy(keys).cuda(), [torch.from_numpy(v).cuda()], dt, keep=None if keep is None else torch.from_numpy(keep).cuda(), key_range=kr)
    _check(sa, dev, want)


@pytest.mark.parametrize("cells,kmin,dtype,keep_some", [(1_000_000, 0, "int64", False), (300_000, -777, "int64", True), (1 << 22, 5, "int64", False),
                                                          ((1 << 22) + 1, 0, "int64", False), (70_000, -35_000, "int32", True), (20_001, 1 << 40, "int64", False)])
def test_narrow_key_ranges_index_the_lds_table_directly(sa, gpu_ready, cells, kmin, dtype, keep_some):
    """round 6: with the key range known and <= 2^22 cells, gb_reduce's table is indexed by the compact record's remainder — the mix is a
    bijection of the range, so (bucket, remainder) is a perfect hash: no keys in the table, no probe, no overflow (`direct_table`).  Same
    results as numpy and as the probing table on the same rows; one cell beyond 2^22 the probing table takes over by itself."""
    rng = np.random.default_rng(cells % 9973)
    n = 2_500_000
    k = (rng.integers(0, cells, n) + kmin).astype(dtype)
    k[:2] = [kmin, kmin + cells - 1]                     # the range's two ends exist
    k[100:50_000] = kmin + cells // 3                    # a key with 2 % of the rows (not peeled: the caller names no heavy key)
    v = rng.normal(3, 2, n)
    v[rng.random(n) < 0.01] = np.nan
    keep = None
    if keep_some:
        keep = (rng.random(n) < 0.7).astype(np.uint8)
        keep[::333] = 2                                   # (not 1: dropped)
    kr = (int(kmin), int(kmin + cells - 1))
    res = sa.groupby_run(k, [v], _DT[dtype], keep=keep, key_range=kr)
    kept = np.ones(n, dtype=bool) if keep is None else keep == 1
    want = _want(k[kept], [v[kept]])
    _check(sa, res, want)
    info = res.info()
    assert info["compact_records"] == 1 and info["retries"] <= 1, info   # (the 2 % key's stream passes its room: ONE exact second attempt; a table never overflows)
    assert info["direct_table"] == (1 if cells <= (1 << 22) else 0), info
    sa.config_set("gb_direct", 0)
    try:
        probing = sa.groupby_run(k, [v], _DT[dtype], keep=keep, key_range=kr)
    finally:
        sa.config_set("gb_direct", 1)
    assert probing.info()["direct_table"] == 0
    _check(sa, probing, want)
    np.testing.assert_array_equal(np.asarray(res.column(sa.GB_KEYS)), np.asarray(probing.column(sa.GB_KEYS)))
    np.testing.assert_array_equal(np.asarray(res.column(sa.GB_COUNT, 0)), np.asarray(probing.column(sa.GB_COUNT, 0)))


def test_dense_ranges_take_the_fused_pass_with_the_direct_table(sa, gpu_ready):
    """Frame.groupby over a dense 1e6-key range (BASELINE configs[3]) on device columns: the fused pass with the direct table answers (one value
    column; several: one such pass per column), the slab-partitioned BinnerOrdinal pass still answers what is outside it (min / max) — same groups either way"""
    import torch
    from vaex_amd.binned import Frame, agg
    rng = np.random.default_rng(77)
    n = 6_000_000
    k = rng.integers(0, 1_000_000, n)
    v = rng.normal(3, 2, n); v[::999] = np.nan
    w = rng.normal(0, 1, n)
    df = Frame(dict(k=torch.from_numpy(k).cuda(), v=torch.from_numpy(v).cuda(), w=torch.from_numpy(w).cuda()), superagg=sa)
    spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v"), "n": agg.count()}
    df.last_groupby_info = None
    got = df.groupby("k", spec)
    info = df.last_groupby_info
    assert info and info.get("direct_table") == 1 and info.get("dense_range_through_fused_pass") == 1, info
    want = _want(k, [v])
    np.testing.assert_array_equal(got["k"], want["k"]); np.testing.assert_array_equal(got["n"], want["rows"]); np.testing.assert_array_equal(got["c"], want["v"][0]["cnt"])
    assert np.all(np.abs(got["s"] - want["v"][0]["s"]) <= 1e-12 * want["v"][0]["sabs"])
    df.dense_through_fused = False                      # the slab-partitioned pair on the same rows
    df.last_groupby_info = None
    slab = df.groupby("k", spec)
    assert not (df.last_groupby_info or {}).get("direct_table")
    for name in ("k", "n", "c"):
        np.testing.assert_array_equal(slab[name], got[name])
    assert np.all(np.abs(slab["s"] - got["s"]) <= 2e-12 * want["v"][0]["sabs"])
    ok = want["v"][0]["cnt"] > 0
    assert np.allclose(np.asarray(slab["m"])[ok], np.asarray(got["m"])[ok], rtol=1e-11, atol=0)
    df.dense_through_fused = True
    df.last_groupby_info = None
    two = df.groupby("k", {"sv": agg.sum("v"), "sw": agg.sum("w"), "hi": agg.max("w")})   # outside the direct form: the dense pass
    assert not (df.last_groupby_info or {}).get("direct_table")
    np.testing.assert_array_equal(two["k"], want["k"])
    assert np.allclose(two["sw"], np.bincount(k, weights=w, minlength=1_000_000)[np.bincount(k, minlength=1_000_000) > 0], rtol=1e-11, atol=1e-9)


@pytest.mark.parametrize("keys", ["dense", "scattered", "wide"])
def test_several_value_columns_are_one_fused_pass_per_column(sa, gpu_ready, keys):
    """round 6: the fused pass's fast forms carry one payload word per record — a call over several value columns is one pass per column where
    the key range leaves compact records (dense: the direct table; scattered: the tag table), one pass per PAIR of columns over wider keys
    (three columns did not ride the fused pass at all before); the passes agree on the groups"""
    import torch
    from vaex_amd.binned import Frame, agg
    rng = np.random.default_rng({"dense": 1, "scattered": 2, "wide": 3}[keys])
    n = 5_000_000
    k = rng.integers(0, 300_000, n)
    if keys == "scattered":
        k = (k * 2654435761) % (1 << 36)
    elif keys == "wide":
        k = (k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).astype(np.int64)
    cols = {c: rng.normal(i, 1 + i, n) for i, c in enumerate("abc")}
    cols["a"][::777] = np.nan
    df = Frame(dict(k=torch.from_numpy(k).cuda(), **{c: torch.from_numpy(x).cuda() for c, x in cols.items()}), superagg=sa)
    spec = {"n": agg.count(), "ca": agg.count("a"), "ma": agg.mean("a"), "sb": agg.sum("b"), "sdb": agg.std("b"), "sc": agg.sum("c"), "mc": agg.mean("c")}
    df.last_groupby_info = None
    got = df.groupby("k", spec)
    info = df.last_groupby_info
    assert info and info.get("passes") == (3 if keys != "wide" else 2) and info.get("value_columns_per_pass") == (1 if keys != "wide" else 2), info
    if keys == "dense":
        assert info.get("direct_table") == 1, info
    want = _want(k, [cols["a"], cols["b"], cols["c"]])
    np.testing.assert_array_equal(got["k"], want["k"]); np.testing.assert_array_equal(got["n"], want["rows"]); np.testing.assert_array_equal(got["ca"], want["v"][0]["cnt"])
    for name, j in (("sb", 1), ("sc", 2)):
        assert np.all(np.abs(got[name] - want["v"][j]["s"]) <= 1e-12 * want["v"][j]["sabs"] + 1e-300), name
    ok = want["v"][0]["cnt"] > 0
    assert np.allclose(np.asarray(got["ma"])[ok], (want["v"][0]["s"] / want["v"][0]["cnt"])[ok], rtol=1e-11, atol=1e-12)
    assert np.allclose(got["mc"], want["v"][2]["s"] / want["v"][2]["cnt"], rtol=1e-11, atol=1e-12)
    # two columns over compact keys: two passes, same numbers as the one 24-byte-record pass they replace
    if keys == "scattered":
        two = df.groupby("k", {"sb": agg.sum("b"), "sc": agg.sum("c")})
        assert df.last_groupby_info.get("passes") == 2
        df.compact_key_bits = 0
        one = df.groupby("k", {"sb": agg.sum("b"), "sc": agg.sum("c")})
        assert not df.last_groupby_info.get("passes")
        np.testing.assert_array_equal(one["k"], two["k"])
        for name, j in (("sb", 1), ("sc", 2)):
            assert np.all(np.abs(one[name] - two[name]) <= 2e-12 * want["v"][j]["sabs"] + 1e-300), name


def test_a_column_overwritten_in_place_is_scanned_again(sa, gpu_ready):
    """the per-column memos (key range, NaN verdict, group count, heavy keys) were keyed on object identity alone — a torch
    tensor overwritten IN PLACE kept them.  They now carry the tensor's version counter."""
    import torch
    from vaex_amd.binned import Frame, agg
    rng = np.random.default_rng(5)
    n = 5_000_000
    k = rng.integers(0, 40_000, n)
    v = rng.normal(0, 1, n)
    kd, vd = torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda()
    df = Frame(dict(k=kd, v=vd), superagg=sa)
    spec = {"c": agg.count("v"), "s": agg.sum("v"), "n": agg.count()}
    first = df.groupby("k", spec)
    np.testing.assert_array_equal(first["k"], np.unique(k))
    kd.mul_(977).add_(-5)                      # another range (dense -> scattered), same object
    vd[::3] = float("nan")                     # NaNs appear in a column that was NaN-free
    torch.cuda.synchronize()
    k2, v2 = k * 977 - 5, v.copy()
    v2[::3] = np.nan
    second = df.groupby("k", spec)
    w = _want(k2, [v2])
    np.testing.assert_array_equal(second["k"], w["k"])
    np.testing.assert_array_equal(second["n"], w["rows"])
    np.testing.assert_array_equal(second["c"], w["v"][0]["cnt"])
    assert np.all(np.abs(second["s"] - w["v"][0]["s"]) <= 1e-12 * w["v"][0]["sabs"])


@pytest.mark.parametrize("dtype,n", [("int64", 5_000_000), ("int32", 3_000_001), ("int16", 400_000), ("uint8", 1_000_000), ("int64", 1000), ("uint32", 2_000_000)])
def test_the_librarys_heavy_key_sample_is_numpys(sa, gpu_ready, dtype, n):
    """vxh_sample_heavy_keys (round 6: the heavy-hitter sample without torch.unique) against numpy on the same strided sample: the keys holding
    >= min_count of 2^17 sampled rows, the 128 most frequent (ties: the smaller key), ascending"""
    import torch
    rng = np.random.default_rng(n % 1000 + len(dtype))
    info = np.iinfo(dtype)
    z = np.minimum(rng.zipf(1.2, n), 50_000)
    if dtype in ("uint8", "int16"):
        k = (z % 200 + (0 if dtype == "uint8" else -100)).astype(dtype)
    elif dtype == "int64":
        k = (z * 2654435761) % (1 << 40) - (1 << 39)
    else:
        k = ((z * 2654435761) % (int(info.max) - int(info.min) + 1) + int(info.min)).astype(dtype)
    tdt = {"uint32": None}.get(dtype, dtype)
    if tdt is None:   # (torch has no uint32 tensors: the same bytes as int32, the library is told the real dtype)
        kd = torch.from_numpy(k.view(np.int32)).cuda()
    else:
        kd = torch.from_numpy(k).cuda()
    m = 1 << 17
    step = max(1, n // m)
    sample = k[::step][:m]
    for share in (1.0 / 1024, 1.0 / 128, 0.5):
        thr = max(8, int(len(sample) * share))
        uniq, cnt = np.unique(sample.astype(np.int64), return_counts=True)
        sel = cnt >= thr
        uniq, cnt = uniq[sel], cnt[sel]
        want = np.sort(uniq[np.argsort(-cnt, kind="stable")[:128]])
        got = np.asarray(sa.sample_heavy_keys(kd, _DT[dtype], m, thr, 128))
        np.testing.assert_array_equal(got, want)
    assert len(np.asarray(sa.sample_heavy_keys(kd, _DT[dtype], m, len(sample) + 1, 128))) == 0


@pytest.mark.parametrize("kind", ["f8", "f4", "i8", "i4", "i2", "i1", "u4", "u2", "u1", "bool"])
def test_code_column_makes_key_codes_and_nan_filled_values(sa, gpu_ready, kind):
    """round 6 (late): vxh_code_column — the int64 codes of a group key with missing values / of a float key (the missing rows under `null_code`, every NaN
    under `nan_code`, float values as the bit patterns of their doubles) and a value column as float64 with NaN where an entry is missing; host and device inputs"""
    import torch
    from vaex_amd.binned import _DT_CODE
    rng = np.random.default_rng(5)
    n = 100_003
    if kind == "bool":
        data = rng.random(n) < 0.4
    elif kind.startswith("f"):
        data = (rng.integers(-40, 40, n) * 0.25).astype(kind)
        data[::17] = np.nan
        data[1::17] = -0.0
    else:
        info = np.iinfo(kind)
        data = rng.integers(info.min, info.max, n, dtype="i8" if kind != "u8" else "u8", endpoint=True).astype(kind)
    mask = rng.random(n) < 0.2
    dt = _DT_CODE[np.dtype(kind).name]
    null_code, nan_code = (1 << 62) + 12345, 0x7ff8000000000000
    want = data.astype("f8").view("i8").copy() if kind.startswith("f") else data.astype("i8")
    if kind.startswith("f"):
        want[np.isnan(data)] = nan_code
    want[mask] = null_code
    wantv = data.astype("f8")
    wantv[mask] = np.nan
    for where in ("host", "device"):
        d = data.view("u1") if kind == "bool" else data
        m = mask.view("u1")
        if where == "device":
            d, m = torch.from_numpy(np.ascontiguousarray(d)).cuda(), torch.from_numpy(np.ascontiguousarray(m)).cuda()
        got = torch.as_tensor(sa.code_column(d, m, dt if kind != "bool" else _DT_CODE["uint8"], 0, null_code, nan_code), device="cuda").cpu().numpy()
        np.testing.assert_array_equal(got, want)
        gotv = torch.as_tensor(sa.code_column(d, m, dt if kind != "bool" else _DT_CODE["uint8"], 1), device="cuda").cpu().numpy()
        np.testing.assert_array_equal(gotv.view("i8")[~np.isnan(wantv)], wantv.view("i8")[~np.isnan(wantv)])
        assert np.array_equal(np.isnan(gotv), np.isnan(wantv))
        nomask = torch.as_tensor(sa.code_column(d, None, dt if kind != "bool" else _DT_CODE["uint8"], 0, null_code, nan_code), device="cuda").cpu().numpy()
        unmasked = data.astype("f8").view("i8").copy() if kind.startswith("f") else data.astype("i8")
        if kind.startswith("f"):
            unmasked[np.isnan(data)] = nan_code
        np.testing.assert_array_equal(nomask, unmasked)


_NP_CMP = [np.less, np.less_equal, np.greater, np.greater_equal, np.equal, np.not_equal]   # vxh_cmp 0..5


def _pred_mask(pred, cols):
    """numpy's answer for a (terms, truth) pair of vxh_groupby_run_selected"""
    terms, truth = pred
    bits = np.zeros(len(cols[0]), dtype=np.uint32)
    with np.errstate(invalid="ignore"):
        for t, (vi, op, c) in enumerate(terms):
            bits |= _NP_CMP[op](cols[vi], np.float64(c)).astype(np.uint32) << t
    return ((truth >> bits) & 1).astype(bool)


@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("keys", ["scattered", "dense", "int32", "heavy"])
def test_groupby_run_selected_terms_over_the_value_columns(sa, gpu_ready, where, keys):
    """vxh_groupby_run_selected (round 6): the filter as terms over the call's own value columns, evaluated by gb_scatter on the payload it
    loads — the groups are those of numpy's mask over the same rows: every comparison, NaN / inf / signed zeros in the values, truth tables
    other than AND, a term per value column, and a keep-mask next to the terms"""
    import torch
    rng = np.random.default_rng({"scattered": 1, "dense": 2, "int32": 3, "heavy": 4}[keys])
    n = 1_200_000
    if keys == "scattered":
        k = (rng.integers(0, 150_000, n) * 2654435761) % (1 << 40) - (1 << 39)
    elif keys == "dense":
        k = rng.integers(0, 300_000, n)
    elif keys == "int32":
        k = rng.integers(-70_000, 70_000, n).astype(np.int32)
    else:
        k = np.where(rng.random(n) < 0.4, 12345, rng.integers(0, 50_000, n)).astype(np.int64)
    v = rng.normal(3, 2, n)
    v[::311] = np.nan; v[11::977] = 0.0; v[13::977] = -0.0; v[17::977] = 3.0; v[7::1013] = 5e-324   # (no infinities among the VALUES: a group's sum would be inf - inf)
    w = rng.normal(0, 1, n); w[::401] = np.nan
    kt = str(k.dtype)
    put = (lambda a: torch.from_numpy(a).cuda()) if where == "device" else (lambda a: a)
    dk, dv, dw = put(k), put(v), put(w)
    kr = (int(k.min()), int(k.max()))
    heavy = np.array([12345], dtype=np.int64) if keys == "heavy" else None
    AND2, OR2, ONLY0_NOT1 = 0b1000, 0b1110, 0b0010
    cases = [([dv], ([(0, op, c)], 0b10)) for op, c in [(0, 3.0), (1, 3.0), (2, 3.0), (3, 3.0), (4, 3.0), (5, 3.0), (4, 0.0), (2, np.inf), (3, -np.inf), (5, np.nan)]]
    cases += [([dv], ([(0, 2, 1.0), (0, 0, 5.0)], AND2)), ([dv], ([(0, 0, 1.0), (0, 2, 5.0)], OR2)), ([dv], ([(0, 2, 1.0)], 0b01)),
              ([dv, dw], ([(0, 2, 3.0), (1, 0, 1.0)], AND2)), ([dv, dw], ([(1, 3, 0.0), (0, 1, 2.5)], ONLY0_NOT1)),
              ([dv, dw], ([(0, 2, 0.0), (1, 2, 0.0), (0, 0, 6.0), (1, 0, 2.0)], 1 << 15))]
    host = {id(dv): v, id(dw): w}
    for values, pred in cases:
        hv = [host[id(a)] for a in values]
        kept = _pred_mask(pred, hv)
        for key_range in (None, kr):
            res = sa.groupby_run(dk, values, _DT[kt], key_range=key_range, heavy=heavy, pred=pred)
            _check(sa, res, _want(k[kept], [a[kept] for a in hv]))
    # a keep-mask next to the terms: both must hold (bytes other than 1 do not keep)
    m = rng.integers(0, 3, n).astype(np.uint8)
    pred = ([(0, 2, 2.0)], 0b10)
    res = sa.groupby_run(dk, [dv], _DT[kt], keep=put(m), key_range=kr, heavy=heavy, pred=pred)
    kept = _pred_mask(pred, [v]) & (m == 1)
    _check(sa, res, _want(k[kept], [v[kept]]))
    # nothing kept: no group
    assert len(sa.groupby_run(dk, [dv], _DT[kt], pred=([(0, 0, -np.inf)], 0b10))) == 0
    with pytest.raises(RuntimeError, match="groupby"):
        sa.groupby_run(dk, [dv], _DT[kt], pred=([(1, 2, 0.0)], 0b10))   # (a term over a value column the call does not have)


def test_frame_groupby_filter_over_its_value_column_is_evaluated_in_the_pass(sa, gpu_ready):
    """Frame.groupby(selection=) whose terms all read the aggregated float64 columns: no mask is made (Frame._mask_array is never asked),
    the result is the keep-mask road's (`groupby_fused_predicate = False`) and numpy's"""
    import torch
    from vaex_amd import binned
    rng = np.random.default_rng(19)
    n = 1_500_000
    k = (rng.integers(0, 200_000, n) * 2654435761) % (1 << 40)
    v = rng.normal(3, 2, n); v[::313] = np.nan
    w = rng.normal(0, 1, n)
    x = rng.normal(0, 1, n)
    spec1 = {"c": binned.agg.count(), "cv": binned.agg.count("v"), "s": binned.agg.sum("v"), "m": binned.agg.mean("v"), "sd": binned.agg.std("v")}
    for device in (False, True):
        cols = dict(k=k, v=v, w=w, x=x)
        if device:
            cols = {c: torch.from_numpy(a).cuda() for c, a in cols.items()}
        for selection, kept in [("(v > 1) & (v < 5)", (v > 1) & (v < 5)), ("v >= 3", v >= 3), ("~(v < 2)", ~(v < 2)), ("(v > 4) | (v < 0)", (v > 4) | (v < 0)), ("v > 2", v > 2)]:
            f = binned.Frame(cols, superagg=sa)
            f.last_groupby_info = None
            asked = []
            real = f._mask_array
            f._mask_array = lambda s, real=real: (asked.append(s), real(s))[1]
            got = f.groupby("k", spec1, selection=selection)
            assert f.last_groupby_info is not None, "the fused pass did not run"
            assert not asked, "a keep-mask was made for a filter over the value column"
            want = _want(k[kept], [v[kept]])
            np.testing.assert_array_equal(got["k"], want["k"]); np.testing.assert_array_equal(got["c"], want["rows"]); np.testing.assert_array_equal(got["cv"], want["v"][0]["cnt"])
            assert np.all(np.abs(got["s"] - want["v"][0]["s"]) <= 1e-12 * want["v"][0]["sabs"])
            plain = binned.Frame(cols, superagg=sa)
            plain.groupby_fused_predicate = False
            ref = plain.groupby("k", spec1, selection=selection)
            for name in got:   # the same rows in the same pass (its LDS atomics add in an order of their own: the float columns agree to rounding)
                if np.asarray(got[name]).dtype.kind == "f":
                    np.testing.assert_allclose(np.asarray(got[name]), np.asarray(ref[name]), rtol=1e-6 if name == "sd" else 1e-10, atol=0, equal_nan=True)   # (std: sum2 / n - mean^2 cancels)
                else:
                    np.testing.assert_array_equal(np.asarray(got[name]), np.asarray(ref[name]))
        # a term over a column that is not aggregated, or an arithmetic term: the keep-mask road as before
        for selection, kept in [("(v > 1) & (x < 0.5)", (v > 1) & (x < 0.5)), ("2 * v + 1 > 6", 2 * v + 1 > 6)]:
            f = binned.Frame(cols, superagg=sa)
            asked = []
            real = f._mask_array
            f._mask_array = lambda s, real=real: (asked.append(s), real(s))[1]
            got = f.groupby("k", spec1, selection=selection)
            assert asked
            want = _want(k[kept], [v[kept]])
            np.testing.assert_array_equal(got["k"], want["k"]); np.testing.assert_array_equal(got["c"], want["rows"])
