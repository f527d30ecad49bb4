"""-m gpu: parity at the REAL shapes of BASELINE.json's configs — configs[1] (2-D 256x256 count+mean), configs[2]
(3-D 128^3 histogram with a uint8 selection) and configs[3] (groupby on a 1e6-cardinality int64 key, dense and
scattered `k*2654435761 % 2**40`, agg sum/mean/std) — on >= 2e8 device-generated rows.  A >= 1e7-row slice of the same
rows is compared with the reference's own C++ (oracle/_ref, the C restatement when that is absent) exactly the way
bench.py's same-run check does: integer grids bit-exact, fp64 sums within 1e-12 * sum|v| of the cell.  The full
size is tied to the slice through linearity: grid(all rows) == grid(slice) + grid(rest).

Also here: the integer cases of the reference's hash-map tests (tests/internal/hash_test.py:81-153, :374-483;
tests/hashmap_unique_test.py) against ordered_set_* / BinnerHash_* on the GPU, and sum-moments 3 and 4
(vaex/agg.py:458-523 skew / kurtosis)."""
import numpy as np
import pytest

from oracle import oracle
from tests import cases

pytestmark = pytest.mark.gpu

N_FULL = 200_000_000
N_SLICE = 10_000_000


@pytest.fixture(autouse=True)
def _ready(sa, gpu_ready):
    for k in ("strategy", "block", "blocks", "parts", "part_chunk"):
        sa.config_set(k, 0)
    sa.config_set("slab_log2", -1)
    yield


def _ref_or_port_case(ref_mod, case):
    """the case on the CPU: the reference's compiled C++ when it loads, else the C restatement"""
    if ref_mod is not None:
        return cases.run_superagg(ref_mod, case, chunk=1 << 20)
    return oracle.run_case(case)


def _ref_module():
    return oracle.ref_module("superagg")


def _sum_tolerance_ok(got, want, absv_sum_per_cell, rtol=1e-12):
    return bool(np.all(np.abs(got - want) <= rtol * absv_sum_per_cell + 0.0))


# ------------------------------------------------------------------------------------------------------------
# configs[1] at full size: bench.py's parity_on_sample as a test
# ------------------------------------------------------------------------------------------------------------
def test_config1_2d_256_count_mean_full_size(sa):
    import torch
    g = torch.Generator(device="cuda").manual_seed(1234)
    n = N_FULL
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    v[::100_003] = float("nan")
    torch.cuda.synchronize()

    def run(lo, hi):
        bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256)
        by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
        grid = sa.Grid([bx, by])
        aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
        bx.set_data(0, x[lo:hi]); by.set_data(0, y[lo:hi])
        aggs[1].set_data(0, v[lo:hi], 0); aggs[2].set_data(0, v[lo:hi], 0)
        grid.bin(0, aggs, hi - lo)
        return [np.array(a.get_result()) for a in aggs], sa.last_kernel(0)

    full, kernel = run(0, n)
    assert kernel.startswith("part_scatter"), kernel  # the bench's kernel pair
    head, _ = run(0, N_SLICE)
    rest, _ = run(N_SLICE, n)
    # linearity over a split of the rows
    np.testing.assert_array_equal(full[0], head[0] + rest[0])
    np.testing.assert_array_equal(full[2], head[2] + rest[2])
    vabs = float(torch.nan_to_num(v).abs().max().item())
    assert _sum_tolerance_ok(full[1], head[1] + rest[1], vabs * np.maximum(full[2], 1))
    assert int(full[0].sum()) == n
    # the slice against the reference
    xs, ys, vs = (t[:N_SLICE].cpu().numpy() for t in (x, y, v))
    case = dict(n=N_SLICE, binners=[dict(kind="scalar", data=xs, vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=ys, vmin=-4, vmax=4, bins=256)],
                aggs=[dict(kind="count"), dict(kind="sum", data=vs), dict(kind="count", data=vs)])
    want = _ref_or_port_case(_ref_module(), case)
    np.testing.assert_array_equal(head[0], want[0])
    np.testing.assert_array_equal(head[2], want[2])
    cases.assert_case_equal(head, want, case)


@pytest.mark.parametrize("box", [True, False])
def test_config1_uniform_data(sa, box):
    """uniform x,y: the densest box holds only ~15 % of the rows — it is still taken (next to part_scatter_blk: staged
    records; the ring-less pass 1 needs ~62 %), and with the box switched off the plain partition pair runs: both must
    agree with the reference"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(99)
    n = 60_000_000
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 8.2 - 4.1
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 8.2 - 4.1
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256)
    by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
    bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
    sa.config_set("hot", 1 if box else 0)
    try:
        grid.bin(0, aggs, n)
        got = [np.array(a.get_result()) for a in aggs]
        if box:
            assert sa.config_get("hot_w") > 0 and 100_000 < sa.config_get("hot_fraction_ppm") < 350_000 and sa.last_kernel(0).startswith("part_scatter_hot")
        else:
            assert sa.config_get("hot_w") == 0
        m = N_SLICE
        for a in aggs:
            a.reset()
        bx.set_data(0, x[:m]); by.set_data(0, y[:m]); aggs[1].set_data(0, v[:m], 0); aggs[2].set_data(0, v[:m], 0)
        grid.bin(0, aggs, m)
        head = [np.array(a.get_result()) for a in aggs]
    finally:
        sa.config_set("hot", 1)
    xs, ys, vs = (t[:m].cpu().numpy() for t in (x, y, v))
    case = dict(n=m, binners=[dict(kind="scalar", data=xs, vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=ys, vmin=-4, vmax=4, bins=256)],
                aggs=[dict(kind="count"), dict(kind="sum", data=vs), dict(kind="count", data=vs)])
    want = _ref_or_port_case(_ref_module(), case)
    cases.assert_case_equal(head, want, case)
    assert int(got[0].sum()) == n


# ------------------------------------------------------------------------------------------------------------
# configs[2]: 3-D 128^3 + boolean selection
# ------------------------------------------------------------------------------------------------------------
def test_config2_3d_128_selection_full_size(sa):
    import torch
    g = torch.Generator(device="cuda").manual_seed(7)
    n = N_FULL
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    z = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    sel = (v > 3).to(torch.uint8)  # the selection vaex materialises as a byte mask (vaex/execution.py:530-549)
    del v
    x[:3] = torch.tensor([float("nan"), float("inf"), -float("inf")], dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()

    def run(lo, hi):
        bs = [sa.BinnerScalar_float64(1, nm, -4.0, 4.0, 128) for nm in "xyz"]
        grid = sa.Grid(bs)
        assert len(grid) == 131 ** 3
        c = sa.AggCount_int64(grid, 1, 1)
        for b, col in zip(bs, (x, y, z)):
            b.set_data(0, col[lo:hi])
        c.set_data_mask(0, sel[lo:hi])
        grid.bin(0, [c], hi - lo)
        return np.array(c.get_result()), sa.last_kernel(0)

    full, kernel = run(0, n)
    assert kernel.startswith("part_scatter"), kernel
    head, _ = run(0, N_SLICE)
    rest, _ = run(N_SLICE, n)
    np.testing.assert_array_equal(full, head + rest)
    assert int(full.sum()) == int(sel.sum().item())
    xs, ys, zs = (t[:N_SLICE].cpu().numpy() for t in (x, y, z))
    case = dict(n=N_SLICE, binners=[dict(kind="scalar", data=c, vmin=-4, vmax=4, bins=128) for c in (xs, ys, zs)],
                aggs=[dict(kind="count", mask=sel[:N_SLICE].cpu().numpy())])
    want = _ref_or_port_case(_ref_module(), case)
    assert want[0].shape == (131, 131, 131)
    np.testing.assert_array_equal(head, want[0])


# ------------------------------------------------------------------------------------------------------------
# configs[3]: groupby on 1e6 int64 keys, sum / mean / std
# ------------------------------------------------------------------------------------------------------------
def _groupby_want(keys, v):
    """per-key count / sum / sum of squares with numpy in double (keys ascending), on a host slice"""
    uniq, inv = np.unique(keys, return_inverse=True)
    ok = v == v
    cnt = np.bincount(inv[ok], minlength=len(uniq))
    s1 = np.bincount(inv[ok], weights=v[ok], minlength=len(uniq))
    s2 = np.bincount(inv[ok], weights=v[ok] * v[ok], minlength=len(uniq))
    sabs = np.bincount(inv[ok], weights=np.abs(v[ok]), minlength=len(uniq))
    return uniq, cnt, s1, s2, sabs


@pytest.mark.parametrize("flavour", ["dense", "scattered"])
def test_config3_groupby_1e6_keys_full_size(sa, flavour):
    import torch
    from vaex_amd.binned import Frame, agg
    g = torch.Generator(device="cuda").manual_seed(11)
    n = N_FULL
    k = torch.randint(0, 1_000_000, (n,), dtype=torch.int64, device="cuda", generator=g)
    if flavour == "scattered":
        k = (k * 2654435761) % (1 << 40)  # forces the hash path
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    v[::77_777] = float("nan")
    torch.cuda.synchronize()
    spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}

    full = Frame(dict(k=k, v=v)).groupby("k", spec)
    assert len(full["k"]) == 1_000_000
    assert np.all(np.diff(full["k"]) > 0)
    assert int(full["c"].sum()) == int((v == v).sum().item())
    head = Frame(dict(k=k[:N_SLICE], v=v[:N_SLICE])).groupby("k", spec)
    rest = Frame(dict(k=k[N_SLICE:], v=v[N_SLICE:])).groupby("k", spec)
    # linearity of count / sum over a split of the rows (keys absent from one part contribute nothing)
    pos_h = np.searchsorted(full["k"], head["k"]); pos_r = np.searchsorted(full["k"], rest["k"])
    c = np.zeros(len(full["k"]), dtype=np.int64); s = np.zeros(len(full["k"]))
    c[pos_h] += head["c"]; c[pos_r] += rest["c"]
    s[pos_h] += head["s"]; s[pos_r] += rest["s"]
    np.testing.assert_array_equal(full["c"], c)
    assert np.all(np.abs(full["s"] - s) <= 1e-12 * 20.0 * np.maximum(full["c"], 1))
    # the slice against numpy in double on the host (the reference path maps keys to ordinals and bins them:
    # same per-key sums; ordinals differ, parity is per key)
    ks, vs = k[:N_SLICE].cpu().numpy(), v[:N_SLICE].cpu().numpy()
    uniq, cnt, s1, s2, sabs = _groupby_want(ks, vs)
    np.testing.assert_array_equal(head["k"], uniq)
    np.testing.assert_array_equal(head["c"], cnt)
    assert np.all(np.abs(head["s"] - s1) <= 1e-12 * sabs)
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = s1 / cnt
        var = s2 / cnt - mean ** 2
    np.testing.assert_allclose(head["m"], mean, rtol=1e-12, equal_nan=True)
    # std: sqrt(m2/n - mean^2) cancels; stated tolerance = 1e-12 of the magnitude that cancels (m2/n), per key
    got_var = head["sd"] ** 2
    okv = cnt > 0
    assert np.all(np.abs(got_var[okv] - var[okv]) <= 4e-12 * (s2[okv] / cnt[okv]) + 1e-300)
    # and against the reference's own C++ through the ordinal binner on codes (what vaex's groupby pass 2 runs)
    ref = _ref_module()
    if ref is not None:
        codes = np.searchsorted(uniq, ks).astype(np.int64)
        case = dict(n=N_SLICE, binners=[dict(kind="ordinal", data=codes, count=len(uniq), min_value=0)],
                    aggs=[dict(kind="sum", data=vs), dict(kind="count", data=vs)])
        want = cases.run_superagg(ref, case, chunk=1 << 20)
        np.testing.assert_array_equal(head["c"], want[1][:len(uniq)])
        assert np.all(np.abs(head["s"] - want[0][:len(uniq)]) <= 1e-12 * sabs)


# ------------------------------------------------------------------------------------------------------------
# sum moments 3 and 4 (skew / kurtosis: vaex/agg.py:458-523) and integer inputs
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("moment", [1, 2, 3, 4])
@pytest.mark.parametrize("strategy", [0, 3, 4])
def test_sum_moments_float64(sa, moment, strategy):
    sa.config_set("strategy", strategy)
    c = cases.gaussian_columns(300_000, seed=5)
    case = dict(n=300_000, binners=[dict(kind="scalar", data=c["x"], vmin=-4, vmax=4, bins=64), dict(kind="scalar", data=c["y"], vmin=-4, vmax=4, bins=64)],
                aggs=[dict(kind="summoment", data=c["v"], moment=moment), dict(kind="count", data=c["v"])])
    want = oracle.run_case(case)
    got = cases.run_superagg(sa, case, to_device=cases.torch_device_array)
    cases.assert_case_equal(got, want, case)
    ref = _ref_module()
    if ref is not None:  # libm pow in the reference vs repeated multiplication here: < 1 ulp per term
        cases.assert_case_equal(got, cases.run_superagg(ref, case), case)
    sa.config_set("strategy", 0)


@pytest.mark.parametrize("dtype", ["int32", "int16", "uint8", "int64", "float32"])
@pytest.mark.parametrize("moment", [2, 3, 4])
def test_sum_moments_other_dtypes(sa, dtype, moment):
    """integer inputs accumulate in int64 / uint64 here; the reference adds pow() to the accumulator in double
    (src/agg_sum.cpp:159: `a += pow(b, moment)` with an integer `a`), so the two agree exactly while every partial
    sum stays below 2^53 — the values below are sized for that (beyond it the reference itself drops low bits in an
    order-dependent way, so there is no single right answer to match)."""
    rng = np.random.default_rng(21)
    n = 200_000
    x = rng.normal(0, 1, n)
    hi = {"int32": 2000, "int16": 300, "uint8": 200, "int64": 3000, "float32": 50}[dtype]
    hi = min(hi, {2: 3000, 3: 300, 4: 60}[moment])  # |sum| stays far below 2^53
    lo = 0 if dtype.startswith("u") else -hi
    v = rng.integers(lo, hi, n).astype(dtype) if dtype != "float32" else rng.normal(0, 10, n).astype("f4")
    case = dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4, vmax=4, bins=32)], aggs=[dict(kind="summoment", data=v, moment=moment), dict(kind="sum", data=v)])
    want = oracle.run_case(case)
    got = cases.run_superagg(sa, case)
    cases.assert_case_equal(got, want, case)
    ref = _ref_module()
    if ref is not None:
        cases.assert_case_equal(got, cases.run_superagg(ref, case), case)


# ------------------------------------------------------------------------------------------------------------
# the GPU hash map: integer cases of tests/internal/hash_test.py and tests/hashmap_unique_test.py
# ------------------------------------------------------------------------------------------------------------
def _check_dense_ordinals(hm, expect_keys):
    keys = np.array(hm.key_array())
    assert len(hm) == len(expect_keys)
    np.testing.assert_array_equal(np.sort(keys), np.sort(np.asarray(expect_keys, dtype=np.int64)))
    ords = np.array(hm.map_ordinal(keys))
    np.testing.assert_array_equal(ords, np.arange(len(keys)))  # key_array is ordered by ordinal; ordinals are dense
    return keys


def test_hashmap_unknown_keys_map_to_minus_one(sa):
    # hash_test.py:391-404 (index_hash: unknown -> -1); ordered_set::map_ordinal src/hash_primitives.hpp:611-691
    hm = sa.ordered_set_int64()
    hm.update(np.array([0, 1, 2], dtype=np.int64))
    np.testing.assert_array_equal(np.sort(hm.map_ordinal(np.array([0, 1, 2], dtype=np.int64))), [0, 1, 2])
    got = hm.map_ordinal(np.array([1, 2, 3, -7, 2**62], dtype=np.int64))
    assert got[2] == -1 and got[3] == -1 and got[4] == -1 and got[0] >= 0 and got[1] >= 0


@pytest.mark.parametrize("dtype", ["int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"])
def test_hashmap_every_integer_dtype(sa, dtype):
    # hash_test.py:69-78 (ordered_set_bool) generalised over the integer key types
    rng = np.random.default_rng(3)
    if dtype == "bool":
        keys = np.array([True, True, False, False, True])
    else:
        info = np.iinfo(dtype)
        keys = rng.integers(max(info.min, -1000), min(info.max, 1000), 5000).astype(dtype)
    hm = getattr(sa, "ordered_set_" + dtype)()
    hm.update(keys)
    uniq = np.unique(keys)
    got = np.array(hm.key_array())
    np.testing.assert_array_equal(np.sort(got), np.sort(uniq.astype(np.int64)))
    ords = np.array(hm.map_ordinal(keys))
    assert ords.min() == 0 and ords.max() == len(uniq) - 1
    np.testing.assert_array_equal(got[ords], keys.astype(np.int64))


def test_hashmap_null_ordinal_and_masked_rows(sa):
    # hash_test.py:81-153 with `missing`: a masked row is the null key, counted once, with its own ordinal (= count)
    keys = np.array([3, 2, 1, 0, 2, 9], dtype=np.int64)
    mask = np.array([0, 0, 1, 0, 0, 1], dtype=np.uint8)
    hm = sa.ordered_set_int64()
    assert hm.null_index == -1 and not hm.has_null
    hm.update(keys, mask)
    assert hm.has_null
    assert len(hm) == 3  # {3, 2, 0}: masked rows add no key
    assert hm.null_index == 3
    got = set(np.array(hm.key_array()).tolist())
    assert got == {3, 2, 0}
    assert hm.map_ordinal(np.array([1, 9], dtype=np.int64)).tolist() == [-1, -1]
    # the null ordinal of the binner: cells [unknown, ordinal 0..N-1, null]
    b = sa.BinnerHash_int64(1, "k", hm)
    grid = sa.Grid([b])
    assert len(grid) == 3 + 2
    c = sa.AggCount_int64(grid, 1, 1)
    probe = np.array([3, 2, 0, 2, 1, 7, 0], dtype=np.int64)
    pmask = np.array([0, 0, 0, 0, 0, 1, 1], dtype=np.uint8)
    b.set_data(0, probe); b.set_data_mask(0, pmask)
    grid.bin(0, [c], len(probe))
    r = np.array(c.get_result())
    ords = hm.map_ordinal(np.array([3, 2, 0], dtype=np.int64))
    assert r[0] == 1            # key 1 unknown (masked at insert time)
    assert r[-1] == 2           # two masked rows -> null cell
    assert r[1 + ords[0]] == 1 and r[1 + ords[1]] == 2 and r[1 + ords[2]] == 1


def test_hashmap_int64_min_key(sa):
    """INT64_MIN is the table's EMPTY sentinel: the key lives in the map's side words and must behave like any other key
    in update / map_ordinal / key_array AND in the binner's probe"""
    lo = np.iinfo(np.int64).min
    keys = np.array([5, lo, 7, lo, 5, np.iinfo(np.int64).max], dtype=np.int64)
    hm = sa.ordered_set_int64()
    hm.update(keys)
    assert len(hm) == 4
    ka = _check_dense_ordinals(hm, [5, lo, 7, np.iinfo(np.int64).max])
    o = hm.map_ordinal(np.array([lo, 6], dtype=np.int64))
    assert o[0] >= 0 and ka[o[0]] == lo and o[1] == -1
    b = sa.BinnerHash_int64(1, "k", hm)
    grid = sa.Grid([b])
    c = sa.AggCount_int64(grid, 1, 1)
    rows = np.array([lo, lo, 5, 6, lo], dtype=np.int64)
    b.set_data(0, rows); b.clear_data_mask(0)
    grid.bin(0, [c], len(rows))
    r = np.array(c.get_result())
    assert r[0] == 1 and r[1 + o[0]] == 3 and r.sum() == 5
    # a map WITHOUT that key: INT64_MIN rows are unknown
    hm2 = sa.ordered_set_int64()
    hm2.update(np.array([1, 2], dtype=np.int64))
    b2 = sa.BinnerHash_int64(1, "k", hm2)
    g2 = sa.Grid([b2])
    c2 = sa.AggCount_int64(g2, 1, 1)
    b2.set_data(0, rows); b2.clear_data_mask(0)
    g2.bin(0, [c2], len(rows))
    assert np.array(c2.get_result())[0] == 5


def test_hashmap_growth_through_two_resizes(sa):
    """the table starts at 2^22 slots; > 2^23 distinct keys take it through two 4x growths (the optimistic insert's
    overflow flag + retry, and the load check before an update) — nothing may be lost or duplicated on the way"""
    import torch
    n = 9_500_000
    base = torch.arange(n, dtype=torch.int64, device="cuda")
    keys = (base * 2654435761) % (1 << 44) - (1 << 43)  # distinct (odd multiplier mod 2^44), scattered, both signs
    assert int(torch.unique(keys).numel()) == n
    torch.cuda.synchronize()
    hm = sa.ordered_set_int64()
    step = 2_500_000
    for i in range(0, n, step):
        part = keys[i:i + step]
        hm.update(torch.cat([part, part[: step // 3]]))  # duplicates inside the update as well
        assert len(hm) == min(n, i + step)
    hm.update(keys)  # everything again: idempotent
    assert len(hm) == n
    ka = np.array(hm.key_array())
    np.testing.assert_array_equal(np.sort(ka), np.sort(keys.cpu().numpy()))
    sample = keys[:: 9973].cpu().numpy()
    ords = np.array(hm.map_ordinal(sample))
    assert ords.min() >= 0 and ords.max() < n
    np.testing.assert_array_equal(ka[ords], sample)
    assert hm.map_ordinal(np.array([(1 << 50) + 1], dtype=np.int64))[0] == -1


def test_hashmap_set_keys_1e6_and_binner(sa):
    # ordered_set::create (src/hash_primitives.hpp:486-537): keys[i] gets ordinal i
    rng = np.random.default_rng(8)
    keys = np.unique(rng.integers(-2**60, 2**60, 1_100_000))[:1_000_000]
    rng.shuffle(keys)
    hm = sa.ordered_set_int64(len(keys))
    hm.set_keys(keys)
    assert len(hm) == 1_000_000
    np.testing.assert_array_equal(np.array(hm.key_array()), keys)
    probe = keys[rng.integers(0, len(keys), 200_000)]
    np.testing.assert_array_equal(np.array(hm.map_ordinal(probe)), _positions(keys, probe))
    with pytest.raises(RuntimeError, match="not empty"):
        hm.set_keys(keys[:3])
    # count rows per key through the binner, against numpy
    rows = np.concatenate([probe, np.array([2**61, -2**61], dtype=np.int64)])
    b = sa.BinnerHash_int64(1, "k", hm)
    grid = sa.Grid([b])
    c = sa.AggCount_int64(grid, 1, 1)
    b.set_data(0, rows); b.clear_data_mask(0)
    grid.bin(0, [c], len(rows))
    r = np.array(c.get_result())
    assert r[0] == 2 and r[-1] == 0
    np.testing.assert_array_equal(r[1:-1], np.bincount(_positions(keys, probe), minlength=len(keys)))


def _positions(keys, probe):
    order = np.argsort(keys)
    return order[np.searchsorted(keys[order], probe)]


def test_hashmap_duplicate_heavy_concurrent_inserts(sa):
    """1e7 rows over 100 distinct keys: every wave races for the same slots; ordinals must stay dense and unique"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    pool = (torch.arange(100, dtype=torch.int64, device="cuda") * 7919 - 333)
    rows = pool[torch.randint(0, 100, (10_000_000,), device="cuda", generator=g)]
    hm = sa.ordered_set_int64()
    hm.update(rows)
    _check_dense_ordinals(hm, pool.cpu().numpy())
    hm.update(rows)
    assert len(hm) == 100


def test_hashmap_unique_matches_numpy_on_random_chunks(sa):
    # tests/hashmap_unique_test.py: keys arrive in chunks (the executor's 1 Mi-row chunks); result = np.unique
    rng = np.random.default_rng(17)
    chunks = [rng.integers(-5000, 5000, 200_000) * 1_000_003 for _ in range(6)]
    hm = sa.ordered_set_int64()
    for ch in chunks:
        hm.update(ch.astype(np.int64))
    allk = np.unique(np.concatenate(chunks))
    _check_dense_ordinals(hm, allk)


@pytest.mark.parametrize("scenario", ["normal", "piled", "hidden_pile", "forced_flush", "queue_overflow", "pile_and_queue_overflow"])
def test_hot_box_packed_counters_are_exact(sa, scenario):
    """round 3: the hot box next to the ring-less pass 1 keeps packed counters — uint16 (10-byte cells: 128x127 instead of 116x115 cells on
    the bench pass) or, where the sampled share of the fullest cell allows, uint8 (9-byte cells: 135x134) that the workgroup flushes
    every few hundred tiles.  Exactness: whenever a workgroup flushes its counters it compares their sum with the hot rows it counted; a
    wrapped counter makes them differ and the call runs again with the next wider counters.
      normal        N(0,1): uint8 counters
      piled         7/8 of the rows in ONE cell, the sample sees it: uint16 straight away, which wraps (114688 rows per workgroup) -> uint32
      hidden_pile   1/8 of the rows in one cell, all of them BETWEEN the sampled segments: uint8 chosen, wraps -> uint16 holds
      forced_flush  a forced box with a flush every 2 trips of the tile loop (the periodic flush itself; a launch of this size would
                    otherwise end before the first one)
      queue_overflow           (round 4) sub-queues forced tiny (`part_cap`): cold records that find their queue full take
                    the slow path — device atomics straight into the grids, which a rerun could not take back.  Next to packed
                    counters that path adds nothing and raises a flag; the call runs again with uint32 counters (one redo)
      pile_and_queue_overflow  the hidden pile (uint8 wraps) AND tiny queues in one call: whichever flag comes first, no row is
                    counted twice
    Every way the result is the uint32 box's and the oracle's."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    n = 1 << 25
    if True:
        piled = scenario == "piled"
        x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
        if piled:
            x[: n - n // 8] = 0.25; y[: n - n // 8] = -0.5   # 7/8 of the rows in one cell: 114688 per workgroup
        if scenario in ("hidden_pile", "pile_and_queue_overflow"):   # (the sample: 8 segments of 2^18 rows starting at multiples of n / 8)
            i = torch.arange(n, device="cuda") % (n // 8)
            hidden = (i >= (1 << 20)) & (i < (1 << 20) + (1 << 19))
            x[hidden] = 0.25; y[hidden] = -0.5              # 1/8 of the rows: 16384 per workgroup in one cell
        bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
        grid = sa.Grid([bx, by])
        aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
        bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
        forced = dict(hot_x0=69, hot_y0=70, hot_w=120, hot_h=120, hot_flush_trips=2) if scenario == "forced_flush" else {}   # (fits next to either pass 1's LDS with uint16 sizing)
        if "queue_overflow" in scenario:
            forced = dict(part_cap=2048)   # records per sub-queue: a handful of the 4096 waves' first blocks fit
        if scenario == "hidden_pile":
            # tiles dealt one by one: every workgroup sees 1/256 of the pile (16384 rows: uint8 wraps, uint16 holds).  With the default
            # super-blocks (wv_span = 16: 32768 consecutive rows per workgroup at a time) the pile's 8 segments land on 32 workgroups,
            # 131072 rows each, and the chain rightly goes on to uint32 — the scenario above (`piled`) already covers that end
            forced = dict(wv_span=1)
        redo0 = sa.config_get("redo_count")
        before = {k: (sa.config_get(k) if k == "wv_span" else 0) for k in forced}   # (config_get of the hot_* keys reports the LAST box, not the forced one: those go back to 0 = not forced)
        for k, val in forced.items():
            sa.config_set(k, val)
        sa.config_set("hot_cache", 0)   # a fresh sample: torch hands the next scenario the previous one's addresses, and the remembered box (and its widest safe counters) with them
        try:
            grid.bin(0, aggs, n)
            used, trips = sa.config_get("hot_cnt16_used"), sa.config_get("hot_flush_trips_used")
        finally:
            sa.config_set("hot_cache", 1)
            for k in forced:
                sa.config_set(k, before[k])
        redone = sa.config_get("redo_count") - redo0
        got = [np.array(a.get_result()) for a in aggs]
        assert sa.last_kernel(0).startswith(("part_scatter_direct_hot", "part_scatter_grouped_hot", "part_scatter_phased_hot")), sa.last_kernel(0)
        assert used == dict(normal=2, piled=0, hidden_pile=1, forced_flush=2, queue_overflow=0, pile_and_queue_overflow=0)[scenario], used   # (what the call ENDED on)
        assert redone == dict(normal=0, hidden_pile=1, forced_flush=0, queue_overflow=1).get(scenario, redone), redone
        if scenario in ("piled", "pile_and_queue_overflow"):   # (uint16 -> uint32, or uint8 -> uint16 -> uint32; both flags at once: uint32 at once)
            assert redone in (1, 2)
        if scenario == "forced_flush":
            assert trips == 2
        elif scenario == "normal":
            assert trips >= 8
        sa.config_set("hot_cnt16", 0)
        try:
            for a in aggs:
                a.reset()
            grid.bin(0, aggs, n)
            want = [np.array(a.get_result()) for a in aggs]
            assert sa.config_get("hot_cnt16_used") == 0
        finally:
            sa.config_set("hot_cnt16", 2)
        np.testing.assert_array_equal(got[0], want[0]); np.testing.assert_array_equal(got[2], want[2])
        assert int(got[0].sum()) == n
        assert np.all(np.abs(got[1] - want[1]) <= 1e-12 * 20.0 * np.maximum(want[0], 1))
        m = N_SLICE
        xs, ys, vs = (t[:m].cpu().numpy() for t in (x, y, v))
        for a in aggs:
            a.reset()
        bx.set_data(0, x[:m]); by.set_data(0, y[:m]); aggs[1].set_data(0, v[:m], 0); aggs[2].set_data(0, v[:m], 0)
        grid.bin(0, aggs, m)
        head = [np.array(a.get_result()) for a in aggs]
        case = dict(n=m, binners=[dict(kind="scalar", data=xs, vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=ys, vmin=-4, vmax=4, bins=256)],
                    aggs=[dict(kind="count"), dict(kind="sum", data=vs), dict(kind="count", data=vs)])
        cases.assert_case_equal(head, _ref_or_port_case(_ref_module(), case), case)


@pytest.mark.parametrize("vdtype", ["int64", "int32", "float32", "f32bin+float64", "f32bin+int64", "f32bin+float32"])
@pytest.mark.parametrize("shape", ["bench_2d", "selection_2d", "uniform_2d", "three_d", "groupby_key"])
def test_other_value_dtypes_ride_the_fast_kernels(sa, shape, vdtype):
    """round 3: value columns that are not float64 next to float64 binners / an int64 key.
    int64 (ids, counts, datetimes: AggSum_int64 / AggCount_int64 into int64 cells) takes part_scatter_wv + the box + part_reduce_fast
    with two's-complement adds in LDS; int32 and float32 columns are converted by part_scatter_wv as it loads them (two 8-byte loads
    per lane) and are int64 / float64 payloads from there on.  Integer sums wrap like the reference's `grid[i] += value` on int64
    (src/agg_sum.cpp:98-127): values up to +-2^62 are in the int64 data.  Integers bit-exact, float32 sums within 1e-12 x sum|v| against
    the reference's C++ on a slice; linear over a split of the rows at the full size."""
    import torch
    f32bin = vdtype.startswith("f32bin+")   # float32 binner columns next to an 8-byte value column: converted on load as well
    vdtype = vdtype.replace("f32bin+", "")
    if f32bin and shape == "groupby_key":
        pytest.skip("an integer key has no float32 form")
    g = torch.Generator(device="cuda").manual_seed(77)
    n = 1 << 26
    if vdtype == "float64":
        v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
        v[::1009] = float("nan")
    elif vdtype == "int64":
        v = torch.randint(-(1 << 40), 1 << 40, (n,), dtype=torch.int64, device="cuda", generator=g)
        v[::1001] = (1 << 62) + 12345
        v[1::1001] = -(1 << 62) - 999
        v[2::5003] = 0x7FF8000000000001   # (the bit pattern of a float64 NaN: an integer like any other)
    elif vdtype == "int32":
        v = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
        v[::1001] = -(1 << 31)
        v[2::5003] = 0x7FC00001           # (the bit pattern of a float32 NaN)
    else:
        v = (torch.randn(n, dtype=torch.float32, device="cuda", generator=g) * 100).contiguous()
        v[::1009] = float("nan")
        v[2::5003] = 1e-42                # (a float32 denormal: widening is exact)
    if shape == "uniform_2d":
        x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
        y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
    else:
        x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    z = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    if f32bin:
        x, y, z = (t.to(torch.float32) for t in (x, y, z))
    Scalar = sa.BinnerScalar_float32 if f32bin else sa.BinnerScalar_float64
    key = torch.randint(0, 200_000, (n,), dtype=torch.int64, device="cuda", generator=g) + 1000
    keep = (torch.rand(n, device="cuda", generator=g) < 0.6).to(torch.uint8)
    torch.cuda.synchronize()
    Sum, Count = getattr(sa, "AggSum_" + vdtype), getattr(sa, "AggCount_" + vdtype)
    floats = vdtype in ("float32", "float64")

    def run(lo, hi):
        if shape == "groupby_key":
            binners = [sa.BinnerOrdinal_int64(1, "k", 200_000, 1000, False, False)]
            binners[0].set_data(0, key[lo:hi])
        elif shape == "three_d":
            binners = [Scalar(1, c, -4.0, 4.0, 64) for c in "xyz"]
            for b, col in zip(binners, (x, y, z)):
                b.set_data(0, col[lo:hi])
        else:
            binners = [Scalar(1, c, -4.0, 4.0, 256) for c in "xy"]
            for b, col in zip(binners, (x, y)):
                b.set_data(0, col[lo:hi])
        grid = sa.Grid(binners)
        aggs = [sa.AggCount_int64(grid, 1, 1), Sum(grid, 1, 1), Count(grid, 1, 1)]
        aggs[1].set_data(0, v[lo:hi], 0); aggs[2].set_data(0, v[lo:hi], 0)
        if shape == "selection_2d":
            for a in aggs:
                a.set_data_mask(0, keep[lo:hi])
        grid.bin(0, aggs, hi - lo)
        return [np.array(a.get_result()) for a in aggs], sa.last_kernel(0)

    full, kernel = run(0, n)
    narrow = vdtype in ("int32", "float32") or f32bin
    # (not the generic pair; a 4-byte column is only converted by part_scatter_wv: where that kernel does not run — 3-d with its 64 slabs,
    #  the 200 000-key groupby — the generic kernels still do)
    if not (narrow and shape in ("three_d", "groupby_key")):
        assert kernel.startswith("part_scatter") and kernel.endswith("_f64" if floats else "_i64"), kernel
    if shape == "bench_2d":
        assert kernel.startswith(("part_scatter_direct_hot", "part_scatter_grouped_hot", "part_scatter_phased_hot")), kernel
    m = 4_000_000
    head, _ = run(0, m)
    rest, _ = run(m, n)
    for k in (0, 2) if floats else (0, 1, 2):
        with np.errstate(over="ignore"):
            np.testing.assert_array_equal(full[k], head[k] + rest[k])
    assert full[1].dtype == (np.float64 if floats else np.int64)
    kept = int(keep.sum().item()) if shape == "selection_2d" else n
    assert int(full[0].sum()) == kept
    if floats:
        assert int(full[2].sum()) == int((~torch.isnan(v) & ((keep == 1) if shape == "selection_2d" else True)).sum().item())
        assert np.all(np.abs(full[1] - (head[1] + rest[1])) <= 1e-12 * 600.0 * np.maximum(full[2], 1))
    else:
        assert int(full[2].sum()) == kept
    cols = dict(x=x, y=y, z=z)
    if shape == "groupby_key":
        bs = [dict(kind="ordinal", data=key[:m].cpu().numpy(), count=200_000, min_value=1000)]
    elif shape == "three_d":
        bs = [dict(kind="scalar", data=cols[c][:m].cpu().numpy(), vmin=-4, vmax=4, bins=64) for c in "xyz"]
    else:
        bs = [dict(kind="scalar", data=cols[c][:m].cpu().numpy(), vmin=-4, vmax=4, bins=256) for c in "xy"]
    vs = v[:m].cpu().numpy()
    aggs = [dict(kind="count"), dict(kind="sum", data=vs), dict(kind="count", data=vs)]
    if shape == "selection_2d":
        ks = keep[:m].cpu().numpy()
        for a in aggs:
            a["mask"] = ks
    case = dict(n=m, binners=bs, aggs=aggs)
    want = _ref_or_port_case(_ref_module(), case)
    if floats:
        np.testing.assert_array_equal(head[0], want[0]); np.testing.assert_array_equal(head[2], want[2])
        cases.assert_case_equal(head, want, case)
    else:
        for k in range(3):
            np.testing.assert_array_equal(head[k], want[k])


@pytest.mark.parametrize("bdtype", ["int64", "int32"])
@pytest.mark.parametrize("shape", ["hours_2d_sum", "ids_1d_count", "three_d_masked", "wide_2d_int64_values"])
def test_integer_binner_columns_ride_the_fast_kernels(sa, shape, bdtype):
    """round 4: int64 / int32 BINNER columns (df.count(binby=[hour, weekday]), ids, datetimes as integers) on grids larger
    than one CU's LDS.  BinnerScalar<T> converts the element to double before `(value - vmin) * scale` (src/binners.cpp:16-35);
    part_scatter_wv does that as it loads the column (PartArgs::bin_ct 2 / 3) instead of the generic pair reading every element through
    its dtype.  Cells bit-exact against the reference's C++ on a slice (integers that sit exactly ON bin edges included), linear over
    a split of the rows at the full size."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(99)
    n = 1 << 26
    tdt = torch.int64 if bdtype == "int64" else torch.int32
    # integer columns whose values land on bin edges of [0, 1024) in 512 bins (every even value IS an edge), below and above the limits
    a = torch.randint(-40, 1100, (n,), dtype=torch.int64, device="cuda", generator=g).to(tdt)
    b = torch.randint(-5, 1030, (n,), dtype=torch.int64, device="cuda", generator=g).to(tdt)
    c = torch.randint(0, 1024, (n,), dtype=torch.int64, device="cuda", generator=g).to(tdt)
    if bdtype == "int64":
        a[::100_003] = (1 << 62)          # (far outside: the overflow cell; exactly representable as a double)
        a[1::100_003] = -(1 << 62)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    v[::1013] = float("nan")
    iv = torch.randint(-(1 << 40), 1 << 40, (n,), dtype=torch.int64, device="cuda", generator=g)
    keep = (torch.rand(n, device="cuda", generator=g) < 0.5).to(torch.uint8)
    torch.cuda.synchronize()
    Scalar = getattr(sa, "BinnerScalar_" + bdtype)
    cols = dict(a=a, b=b, c=c)
    # (grids of <= 16 slabs: the box-less part_scatter_wv keeps a ring per (wave, slab) — beyond that its rings do not fit the LDS and the
    #  generic pair still serves integer binner columns)
    spec = {"hours_2d_sum": (["a", "b"], 256, "f64"), "ids_1d_count": (["a"], 1 << 17, None), "three_d_masked": (["a", "b", "c"], 32, "f64"),
            "wide_2d_int64_values": (["b", "c"], 384, "i64")}[shape]
    names, bins, vkind = spec
    lim = (0.0, float(1 << 17)) if shape == "ids_1d_count" else (0.0, 1024.0)

    def run(lo, hi):
        binners = [Scalar(1, nm, lim[0], lim[1], bins) for nm in names]
        for bn, nm in zip(binners, names):
            bn.set_data(0, cols[nm][lo:hi])
        grid = sa.Grid(binners)
        if vkind == "f64":
            aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
            aggs[1].set_data(0, v[lo:hi], 0); aggs[2].set_data(0, v[lo:hi], 0)
        elif vkind == "i64":
            aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_int64(grid, 1, 1)]
            aggs[1].set_data(0, iv[lo:hi], 0)
        else:
            aggs = [sa.AggCount_int64(grid, 1, 1)]
        if shape == "three_d_masked":
            for ag in aggs:
                ag.set_data_mask(0, keep[lo:hi])
        grid.bin(0, aggs, hi - lo)
        return [np.array(ag.get_result()) for ag in aggs], sa.last_kernel(0)

    full, kernel = run(0, n)
    assert kernel.startswith("part_scatter_wv") and "generic" not in kernel, kernel
    m = 4_000_000
    head, _ = run(0, m)
    rest, _ = run(m, n)
    np.testing.assert_array_equal(full[0], head[0] + rest[0])
    assert int(full[0].sum()) == (int(keep.sum().item()) if shape == "three_d_masked" else n)
    if vkind == "i64":
        with np.errstate(over="ignore"):
            np.testing.assert_array_equal(full[1], head[1] + rest[1])
    bs = [dict(kind="scalar", data=cols[nm][:m].cpu().numpy(), vmin=lim[0], vmax=lim[1], bins=bins) for nm in names]
    if vkind == "f64":
        vs = v[:m].cpu().numpy()
        aggs = [dict(kind="count"), dict(kind="sum", data=vs), dict(kind="count", data=vs)]
    elif vkind == "i64":
        aggs = [dict(kind="count"), dict(kind="sum", data=iv[:m].cpu().numpy())]
    else:
        aggs = [dict(kind="count")]
    if shape == "three_d_masked":
        ks = keep[:m].cpu().numpy()
        for ag in aggs:
            ag["mask"] = ks
    case = dict(n=m, binners=bs, aggs=aggs)
    want = _ref_or_port_case(_ref_module(), case)
    np.testing.assert_array_equal(head[0], want[0])
    if vkind == "i64":
        np.testing.assert_array_equal(head[1], want[1])
    elif vkind == "f64":
        np.testing.assert_array_equal(head[2], want[2])
        cases.assert_case_equal(head, want, case)


# ------------------------------------------------------------------------------------------------------------
# round 4: the GROUPED pass 1 ("wv" = 5: cold records compacted into a wave-private ring, slab-sorted 64-record groups in one
# stream per wave, pass 2 = part_reduce_grp) against the ring-less one ("wv" = 3) and the reference's C++
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("form", [5, 6])
@pytest.mark.parametrize("variant", ["bench", "masked", "count_only", "std", "int64_values", "tiny_blocks", "region_overflow", "nan_values", "sigma2"])
def test_grouped_pass1(sa, variant, form):
    """form 5: the grouped pass 1; form 6 (round 5): the same with the groups' record stores held back in a register queue and issued in
    chip-wide bursts on the flips of a wall-clock bit ("wv" = 6, part_scatter_wv<..., DIRECT = 4>; sigma2 fills the queue between two
    flips, tiny_blocks leaves a block with groups still held, region_overflow meets the slow path with groups held).
    Every variant bins the same rows with wv = 5 / 6 and wv = 3: integer grids bit-exact, fp64 sums within 1e-12 x sum|v| of the cell, and
    a 1e7-row slice equals the reference's C++ (restatement when absent).
      bench            256x256 count(*) + sum(v) + count(v), N(0,1) x,y, an odd row count (partial tile, partial last group)
      masked           ... one keep-mask shared by the three aggregators
      count_only       count(*) alone (records without a value)
      std              + the sum of squares (three-plane box, uint32 counters)
      int64_values     integer sums (payload bits pass through)
      tiny_blocks      one group per reserved block: every flush reserves (the in-line reservation path)
      region_overflow  regions forced tiny: the slow path next to packed counters -> rerun with uint32 counters
      nan_values       1 % NaN in v: cold by definition (the box takes non-NaN values only)
      sigma2           x,y ~ N(0, 2): ~40 % of the rows cold, many groups per wave"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(31)
    n = (1 << 25) + 12_345
    scale = 2.0 if variant == "sigma2" else 1.0
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * scale
    y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * scale
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    if variant == "nan_values":
        v[torch.rand(n, device="cuda", generator=g) < 0.01] = float("nan")
    if variant == "int64_values":
        v = (v * 1000).to(torch.int64)
    keep = (torch.rand(n, device="cuda", generator=g) < 0.6).to(torch.uint8) if variant == "masked" else None
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    if variant == "count_only":
        aggs = [sa.AggCount_int64(grid, 1, 1)]
    elif variant == "int64_values":
        aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_int64(grid, 1, 1), sa.AggCount_int64(grid, 1, 1)]
    else:
        aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
        if variant == "std":
            aggs.append(sa.AggSumMoment_float64(grid, 1, 1, 2))

    def run(rows, wv, **knobs):
        for a in aggs:
            a.reset()
        bx.set_data(0, x[:rows]); by.set_data(0, y[:rows])
        for a in aggs[1:]:
            a.set_data(0, v[:rows], 0)
        for a in aggs:
            if keep is not None:
                a.set_data_mask(0, keep[:rows])
            else:
                a.clear_data_mask(0)
        sa.config_set("wv", wv)
        sa.config_set("hot_cache", 0)
        for k, val in knobs.items():
            sa.config_set(k, val)
        try:
            grid.bin(0, aggs, rows)
            kernel = sa.last_kernel(0)
        finally:
            sa.config_set("wv", 6); sa.config_set("hot_cache", 1)   # (the library's default)
            for k in knobs:
                sa.config_set(k, 62 if k == "hot_direct_pct" else (12 if k == "wv_phase" else 0))
        return [np.array(a.get_result()) for a in aggs], kernel

    knobs = dict(tiny_blocks=dict(wv_block=64), region_overflow=dict(part_cap=4096), sigma2=dict(hot_direct_pct=20), count_only=dict(strategy=4)).get(variant, {})   # (sigma2: the ring-less family down to a 20 % box, where the default hands over to part_scatter_blk at 62 %)
    redo0 = sa.config_get("redo_count")
    got, kernel = run(n, form, **knobs)
    redone = sa.config_get("redo_count") - redo0
    if sa.config_get("last_slabs") > 8:   # (the groups' header holds eight slabs: wider signatures keep the per-(wave, slab) streams)
        assert kernel.startswith("part_scatter_direct_hot"), kernel
        pytest.skip(f"{variant}: {sa.config_get('last_slabs')} slabs, the grouped layout serves <= 8")
    assert kernel.startswith("part_scatter_grouped_hot" if form == 5 else "part_scatter_phased_hot"), kernel
    if form == 6 and variant in ("bench", "masked"):   # the other end of the phase knob: a flip every 160 ns — every tile ends in a burst
        again, _ = run(n, form, wv_phase=4, **knobs)
        for a, b in zip(again, got):
            assert np.array_equal(a, b) if a.dtype.kind in "iu" else np.all(np.abs(a - b) <= 1e-12 * 20.0 * np.maximum(got[0], 1)), variant
    if variant == "region_overflow":
        assert redone >= 1 or sa.config_get("hot_cnt16_used") == 0
    want, kernel3 = run(n, 3, **{k: val for k, val in knobs.items() if k in ("hot_direct_pct", "strategy")})
    assert kernel3.startswith("part_scatter_direct_hot"), kernel3
    vabs = torch.nan_to_num(v.to(torch.float64)).abs().max().item()
    for k, (a, b) in enumerate(zip(got, want)):
        if a.dtype.kind in "iu":
            np.testing.assert_array_equal(a, b, err_msg=f"{variant}: aggregator {k}")
        else:
            scale_k = vabs ** 2 if k == 3 else vabs
            assert np.all(np.abs(a - b) <= 1e-12 * scale_k * np.maximum(got[0], 1)), f"{variant}: aggregator {k}"
    expect_rows = int(keep.sum().item()) if keep is not None else n
    assert int(got[0].sum()) == expect_rows
    m = N_SLICE
    head, _ = run(m, form, **knobs)
    xs, ys = x[:m].cpu().numpy(), y[:m].cpu().numpy()
    vs = v[:m].cpu().numpy()
    ks = None if keep is None else keep[:m].cpu().numpy().astype(bool)
    mk = lambda d: dict(d, mask=ks) if ks is not None else d
    if variant == "count_only":
        ca = [mk(dict(kind="count"))]
    else:
        ca = [mk(dict(kind="count")), mk(dict(kind="sum", data=vs)), mk(dict(kind="count", data=vs))]
        if variant == "std":
            ca.append(mk(dict(kind="summoment", data=vs, moment=2)))
    case = dict(n=m, binners=[dict(kind="scalar", data=xs, vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=ys, vmin=-4, vmax=4, bins=256)], aggs=ca)
    cases.assert_case_equal(head, _ref_or_port_case(_ref_module(), case), case)


def test_default_pass1_next_to_a_box_is_the_phased_grouped_form(sa):
    """round 5: without any knob the bench signature runs part_scatter_wv<..., DIRECT = 4> (cold records in slab-sorted groups, held in
    registers, written in chip-wide bursts) + part_reduce_grp; the timed trial between the grouped and the ring-less form of round 4 is
    gone (the phased form is ahead of both on every box: profiles/r05_headline_ab.txt).  Same grids as the ring-less form."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(3)
    n = (1 << 26) + 999
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
    bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
    assert sa.config_get("wv") == 6 and sa.config_get("wv_phase") == 12
    kernels, results = [], []
    try:
        for wv in (6, 6, 3):
            sa.config_set("wv", wv)
            for a in aggs:
                a.reset()
            grid.bin(0, aggs, n)
            kernels.append(sa.last_kernel(0))
            results.append([np.array(a.get_result()) for a in aggs])
    finally:
        sa.config_set("wv", 6)
    assert kernels[0] == kernels[1] == "part_scatter_phased_hot+part_reduce_grp_f64" and kernels[2].startswith("part_scatter_direct_hot"), kernels
    for r in results[1:]:
        assert np.array_equal(r[0], results[0][0]) and np.array_equal(r[2], results[0][2])
        assert np.all(np.abs(r[1] - results[0][1]) <= 1e-12 * 20.0 * np.maximum(results[0][0], 1))
    assert int(results[0][0].sum()) == n



def test_count_256x256_takes_the_box_when_the_sample_says_so(sa):
    """round 5: count(*) on 256 x 256 (north_star's target sentence) fits one workgroup's LDS only with packed uint16 counters, i.e. a
    returning LDS atomic per row.  Where a sample of the columns says the hot box of the partition strategy (194 x 194 uint32 counters)
    holds >= 90 % of the rows, the call takes that road (phased pass 1 + part_reduce_grp); spread-out data and the knob set to 0 keep
    count_lds_f64<PACK16>.  Same grid either way, every row counted, with and without a keep-mask."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(19)
    n = (1 << 26) + 777
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    xu = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
    keep = (torch.rand(n, device="cuda", generator=g) < 0.5).to(torch.uint8)

    def count(cx, cy, mask=None, **knobs):
        bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
        grid = sa.Grid([bx, by])
        c = sa.AggCount_int64(grid, 1, 1)
        bx.set_data(0, cx); by.set_data(0, cy)
        if mask is not None:
            c.set_data_mask(0, mask)
        for k, val in knobs.items():
            sa.config_set(k, val)
        try:
            grid.bin(0, [c], n)
            return np.array(c.get_result()), sa.last_kernel(0)
        finally:
            for k in knobs:
                sa.config_set(k, 90 if k == "count_box_pct" else 0)

    via_box, k_box = count(x, y)
    packed, k_packed = count(x, y, count_box_pct=0)
    assert k_box.startswith("part_scatter_phased_hot") and k_packed.startswith(("count_lds16", "bin_lds_count16")), (k_box, k_packed)
    np.testing.assert_array_equal(via_box, packed)
    assert int(via_box.sum()) == n
    yu = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
    wide, k_wide = count(xu, y)                           # uniform x, N(0,1) y: the box search turns the box into a 259 x 145 band — still > 90 %
    assert k_wide.startswith("part_scatter_phased_hot") and int(wide.sum()) == n, k_wide
    spread, k_spread = count(xu, yu)                      # uniform x AND y: no box holds more than ~57 % of the sample — the packed kernel stays
    assert k_spread.startswith(("count_lds16", "bin_lds_count16")), k_spread
    assert int(spread.sum()) == n
    m_box, k_mbox = count(x, y, keep)
    m_packed, _ = count(x, y, keep, count_box_pct=0)
    np.testing.assert_array_equal(m_box, m_packed)
    assert int(m_box.sum()) == int(keep.sum().item())


# ------------------------------------------------------------------------------------------------------------
# round 5: scalar binner columns of dtypes the fast kernels do not read — int8 / int16 / unsigned / bool,
# byte-swapped, or with a missing-value mask — are converted to float64 by a pass of their own ("convert_binners": calls of >= 2^22
# device rows) and then ride the float64 fast paths; BinnerScalar<T> converts to double first anyway (src/binners.cpp:16-35) and a
# masked row lands where a NaN lands (cell 0), so the grids must be the reference's bit for bit.
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["int8", "int16", "uint8", "uint16", "uint32", "uint64", "bool", "float64_be", "int32_be", "float32_be",
                                  "float64_masked", "int16_masked", "int64_be_masked"])
def test_other_binner_dtypes_are_converted_for_the_fast_kernels(sa, kind):
    rng = np.random.default_rng(abs(hash(kind)) % 1000)
    n = (1 << 23) + 1_001
    base = kind.split("_")[0]
    be, masked = "_be" in kind, "masked" in kind
    if base == "bool":
        a, b, lim = rng.random(n) < 0.3, rng.random(n) < 0.6, (0.0, 2.0)
    elif base.startswith("float"):
        a, b, lim = rng.normal(0, 1, n).astype(base), rng.normal(0, 1, n).astype(base), (-4.0, 4.0)
        a[::977] = np.nan
    else:
        info = np.iinfo(base)
        lo, hi = max(info.min, -120), min(int(info.max), 1000)
        a, b = rng.integers(lo, hi, n).astype(base), rng.integers(lo, hi, n).astype(base)
        if base in ("uint64", "int64"):
            a[::50_021] = info.max            # (beyond 2^53: double(value) rounds, as BinnerScalar<T> does)
        lim = (float(lo) + 0.5, float(hi) * 0.75)
    if be:
        a, b = a.astype(a.dtype.newbyteorder(">")), b.astype(b.dtype.newbyteorder(">"))
    ma = (rng.random(n) < 0.1) if masked else None
    v = rng.normal(3, 2, n); v[::1013] = np.nan
    shape = 64 if base == "bool" else 256
    case = dict(n=n, binners=[dict(kind="scalar", data=a, vmin=lim[0], vmax=lim[1], bins=shape, mask=ma), dict(kind="scalar", data=b, vmin=lim[0], vmax=lim[1], bins=shape)],
                aggs=[dict(kind="count"), dict(kind="sum", data=v), dict(kind="count", data=v)])
    c0 = sa.config_get("converted_calls")
    got = cases.run_superagg(sa, case, to_device=cases.torch_device_array)
    kernel = sa.last_kernel(0)
    assert sa.config_get("converted_calls") == c0 + 1, kind
    assert "generic" not in kernel, kernel
    sa.config_set("convert_binners", 0)
    try:
        generic = cases.run_superagg(sa, case, to_device=cases.torch_device_array)
        assert sa.config_get("converted_calls") == c0 + 1
    finally:
        sa.config_set("convert_binners", 1 << 22)
    np.testing.assert_array_equal(got[0], generic[0]); np.testing.assert_array_equal(got[2], generic[2])
    assert int(got[0].sum()) == n
    m = 2_000_000
    head_case = dict(n=m, binners=[dict(bd, data=bd["data"][:m], mask=None if bd.get("mask") is None else bd["mask"][:m]) for bd in case["binners"]],
                     aggs=[dict(ad, data=None if ad.get("data") is None else ad["data"][:m]) for ad in case["aggs"]])
    # (a slice below the conversion threshold: the generic kernels — pinned against the reference's C++ elsewhere — and the converted
    #  full-size call must agree with the reference on it through linearity: head + rest = whole)
    want = _ref_or_port_case(_ref_module(), head_case)
    sa.config_set("convert_binners", 1)
    try:
        head = cases.run_superagg(sa, head_case, to_device=cases.torch_device_array)
    finally:
        sa.config_set("convert_binners", 1 << 22)
    cases.assert_case_equal(head, want, head_case)
    cases.assert_case_equal(got, generic, case)
    # ONE value column (the harness above uploads v once per aggregator = two): 8- / 16-bit integers, bool and float32 columns are
    # converted to FLOAT32 (exact for them, half the bytes) and ride the float32-binner fast paths
    one = dict(case, aggs=case["aggs"][:2])
    got1 = cases.run_superagg(sa, one, to_device=cases.torch_device_array)
    assert "generic" not in sa.last_kernel(0), sa.last_kernel(0)
    np.testing.assert_array_equal(got1[0], got[0])
    cases.assert_case_equal(got1, generic[:2], one)


@pytest.mark.parametrize("kind", ["int8", "int16", "uint8", "uint16", "uint32", "uint64", "bool", "int16_be", "int32_be", "int64_be", "uint32_be", "float64_be", "float32_be"])
def test_other_value_dtypes_are_converted_for_the_fast_kernels(sa, kind):
    """round 6 (VERDICT r5 missing #4): count / sum over a value column of a dtype the typed paths do not load — int8 / int16 / unsigned / bool /
    byte-swapped — is converted by one pass into int64 (float64) and rides them (upcast<T>, src/agg_sum.cpp:6-62, gives such a column the int64 /
    uint64 / float64 grid those paths fill): the same grids as the generic kernels, integer sums bit for bit, and as the reference's C++ on a slice"""
    rng = np.random.default_rng(abs(hash(kind)) % 1000 + 7)
    n = (1 << 23) + 1_001
    base, be = kind.split("_")[0], "_be" in kind
    if base == "bool":
        v = rng.random(n) < 0.3
    elif base.startswith("float"):
        v = rng.normal(3, 2, n).astype(base); v[::1013] = np.nan
    else:
        info = np.iinfo(base)
        v = rng.integers(max(info.min, -30_000), min(int(info.max), 60_000), n).astype(base)
        if base in ("uint64", "int64"):
            v[::50_021] = info.max // 4          # (far beyond 2^53: the sums are integer sums)
    if be:
        v = v.astype(v.dtype.newbyteorder(">"))
    x, y = rng.normal(0, 1, n), rng.normal(0, 1, n)
    x[::977] = np.nan
    case = dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4.0, vmax=4.0, bins=256), dict(kind="scalar", data=y, vmin=-4.0, vmax=4.0, bins=256)],
                aggs=[dict(kind="count"), dict(kind="sum", data=v), dict(kind="count", data=v)])
    c0 = sa.config_get("converted_value_calls")
    got = cases.run_superagg(sa, case, to_device=cases.torch_device_array)
    kernel = sa.last_kernel(0)
    conv = 0 if kind == "uint64" else 1          # (a native uint64 column needs no pass: its 64-bit adds ARE the int64 path's)
    assert sa.config_get("converted_value_calls") == c0 + conv, kind
    assert "generic" not in kernel, kernel
    sa.config_set("convert_binners", 0)
    try:
        generic = cases.run_superagg(sa, case, to_device=cases.torch_device_array)
        assert sa.config_get("converted_value_calls") == c0 + conv and (("generic" in sa.last_kernel(0)) == bool(conv)), sa.last_kernel(0)
    finally:
        sa.config_set("convert_binners", 1 << 22)
    np.testing.assert_array_equal(got[0], generic[0]); np.testing.assert_array_equal(got[2], generic[2])
    assert got[1].dtype == generic[1].dtype
    if not base.startswith("float"):
        np.testing.assert_array_equal(got[1], generic[1])        # integer sums: exact
    cases.assert_case_equal(got, generic, case)
    assert int(got[0].sum()) == n
    m = 1_500_000
    head_case = dict(n=m, binners=[dict(bd, data=bd["data"][:m]) for bd in case["binners"]], aggs=[dict(ad, data=None if ad.get("data") is None else ad["data"][:m]) for ad in case["aggs"]])
    want = _ref_or_port_case(_ref_module(), head_case)
    sa.config_set("convert_binners", 1)
    try:
        head = cases.run_superagg(sa, head_case, to_device=cases.torch_device_array)
        assert sa.config_get("converted_value_calls") == c0 + 2 * conv
    finally:
        sa.config_set("convert_binners", 1 << 22)
    cases.assert_case_equal(head, want, head_case)
    # host chunks take the same road (staged, then converted on the device)
    host = cases.run_superagg(sa, case, chunk=1 << 23, nthreads=1)
    np.testing.assert_array_equal(host[0], got[0])
    if not base.startswith("float"):
        np.testing.assert_array_equal(host[1], got[1])
