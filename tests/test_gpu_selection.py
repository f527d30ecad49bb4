"""Device-side selections (include/vaex_hip.h "device-side selections", vaex_amd/csrc/vxh_select.hip): a
selection handed over as a predicate and evaluated on the GPU must keep exactly the rows numpy keeps when it evaluates the
same expression on the host (what vaex does per chunk: vaex/execution.py:530-549) — compared through the same aggregation
fed with the numpy-built mask, and against plain numpy for the counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sa = pytest.importorskip("vaex_amd.superagg")
from vaex_amd import predicate as P  # noqa: E402
from vaex_amd.binned import Frame  # noqa: E402

LIM = [[-4, 4], [-4, 4]]


def _columns(n, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, n)
    x[::501] = np.nan
    cols = dict(
        x=x, y=rng.normal(0, 1, n), v=rng.normal(3, 2, n),
        f=rng.normal(0, 1, n).astype("f4"),
        i=rng.integers(-(1 << 62), 1 << 62, n),
        j=rng.integers(-100, 100, n).astype("i4"), h=rng.integers(-100, 100, n).astype("i2"), b=rng.integers(-100, 100, n).astype("i1"),
        U=rng.integers(0, 1 << 63, n).astype("u8") * 2, u=rng.integers(0, 1 << 32, n).astype("u4"), w=rng.integers(0, 1 << 16, n).astype("u2"),
        c=rng.integers(0, 256, n).astype("u1"), t=rng.random(n) < 0.5,
    )
    cols["v"][::333] = np.nan
    return cols


EXPRS = [
    "v > 3", "(x > 0) & (v < 3.5)", "~(x > 0) | (y >= 1)", "(-1 < y) & (y <= 1)", "x != 0.25", "f < 0.5", "(f >= -1) & (f <= 1) & (j != 0) & (h > -50)",
    "i > 0", "i >= 4611686018427387000", "i > 0.5", "U >= 9223372036854775807", "U > -1", "U < -1", "(u < 2147483648) | (w == 17)", "b <= -3",
    "c != 255", "t == 1", "(t != 0) & (c > 10)", "j > 2.5",
]


def _want_mask(expr, cols):
    p = P.compile_selection(expr, cols)
    return p.numpy_mask(cols)


@pytest.mark.parametrize("n,chunk,threads", [(200_003, 50_000, 3), (3_000_000, 1 << 20, 4)])
def test_predicates_keep_the_rows_numpy_keeps(n, chunk, threads):
    cols = _columns(n, 1)
    f = Frame(cols, chunk_size=chunk, nthreads=threads)
    for expr in EXPRS:
        keep = _want_mask(expr, cols)
        got = f.count(binby=["x", "y"], limits=LIM, shape=64, selection=expr, edges=True)
        want = f.count(binby=["x", "y"], limits=LIM, shape=64, selection=keep, edges=True)
        assert np.array_equal(got, want), expr
        assert int(got.sum()) == int(keep.sum()), expr


def test_python_and_numpy_agree_on_the_expressions_used_here():
    cols = _columns(10_000, 2)
    with np.errstate(invalid="ignore"):
        for expr in EXPRS:
            want = eval(expr, {}, dict(cols))
            assert np.array_equal(_want_mask(expr, cols), want), expr


def test_sums_and_moments_with_a_predicate_and_missing_values():
    n = 1_500_000
    cols = _columns(n, 3)
    miss = np.random.default_rng(4).random(n) < 0.1
    cols["m"] = np.ma.array(cols["y"] * 10, mask=miss)
    f = Frame(cols, chunk_size=1 << 19, nthreads=2)
    expr = "(x > -1) & (v < 6)"
    keep = _want_mask(expr, cols)
    for method in ("sum", "mean", "std", "min", "max"):
        got = getattr(f, method)("m", binby=["x", "y"], limits=LIM, shape=32, selection=expr)
        want = getattr(f, method)("m", binby=["x", "y"], limits=LIM, shape=32, selection=keep)
        if method in ("min", "max"):
            assert np.array_equal(got, want, equal_nan=True), method
        else:  # fp64 sums in a different order (chunks land on different slots): 1e-12 of the summed magnitude
            scale = np.nanmax(np.abs(want)) if method == "sum" else 1.0
            assert np.allclose(got, want, rtol=1e-9 if method == "std" else 1e-12, atol=1e-12 * scale, equal_nan=True), method


def test_device_resident_columns_and_several_selections_in_one_pass():
    import torch
    from vaex_amd import binned
    n = 4_000_000
    cols = _columns(n, 5)
    host = {k: cols[k] for k in ("x", "y", "v", "j")}
    dev = {k: torch.from_numpy(a).cuda() for k, a in host.items()}
    torch.cuda.synchronize()
    f = Frame(dev)
    descs = [binned.agg.count(selection="v > 3"), binned.agg.mean("v", selection="v > 3"), binned.agg.count(selection="(j >= 0) & (x < 0.5)"), binned.agg.sum("v")]
    got = f._agg(descs, ["x", "y"], LIM, 256)
    k1, k2 = _want_mask("v > 3", host), _want_mask("(j >= 0) & (x < 0.5)", host)
    fh = Frame(host, nthreads=2)
    want = fh._agg([binned.agg.count(selection=k1), binned.agg.mean("v", selection=k1), binned.agg.count(selection=k2), binned.agg.sum("v")], ["x", "y"], LIM, 256)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert np.allclose(got[1], want[1], rtol=1e-12, atol=0, equal_nan=True) and np.allclose(got[3], want[3], rtol=1e-12, atol=0, equal_nan=True)
    assert int(got[0].sum()) == int((k1 & (np.abs(host["x"]) < 4) & (np.abs(host["y"]) < 4)).sum())
    # a 3-D pass with the selection over a binned column (the partition strategy's masked path)
    z = torch.from_numpy(cols["f"].astype("f8")).cuda()
    f3 = Frame(dict(dev, z=z))
    lim3 = LIM + [[-4, 4]]
    g = f3.count(binby=["x", "y", "z"], limits=lim3, shape=64, selection="x > 0.125")
    w = Frame(dict(host, z=cols["f"].astype("f8")), nthreads=2).count(binby=["x", "y", "z"], limits=lim3, shape=64, selection=_want_mask("x > 0.125", host))
    assert np.array_equal(g, w)


def test_c_abi_argument_checks():
    with pytest.raises(RuntimeError, match="1 to 4 terms"):
        sa.Selection(1, [0], [], 0)
    with pytest.raises(RuntimeError, match="column that does not exist"):
        sa.Selection(1, [0], [(1, sa.CMP_GT, 0.0)], 2)
    sel = sa.Selection(1, [0], [(0, sa.CMP_GT, 0.0)], 2)
    with pytest.raises(RuntimeError, match="Itemsize"):
        sel.set_data(0, 0, np.zeros(4, dtype="f4"))
    b = sa.BinnerScalar_float64(1, "x", 0.0, 1.0, 4)
    g = sa.Grid([b])
    a = sa.AggCount_float64(g, 1, 1)
    x = np.linspace(-1, 2, 100)
    b.set_data(0, x); a.set_data(0, x, 0); a.set_selection(sel)
    with pytest.raises(RuntimeError, match="selection data not set"):
        g.bin(0, [a], len(x))
    sel.set_data(0, 0, x[:50])
    with pytest.raises(RuntimeError, match="selection data is shorter"):
        g.bin(0, [a], len(x))
    sel.set_data(0, 0, x)
    g.bin(0, [a], len(x))
    r = np.asarray(a.get_result())
    assert r[2:-1].sum() == ((x > 0) & (x >= 0) & (x < 1)).sum()
    a.set_selection(None)
    g.bin(0, [a], len(x))
    assert np.asarray(a.get_result())[2:-1].sum() == ((x > 0) & (x < 1)).sum() + ((x >= 0) & (x < 1)).sum()


def test_float32_columns_are_compared_in_float32_like_numpy():
    """numpy compares a float32 column with a Python float in float32: `f4 <= 0.3` holds for float32(0.3) (0.30000001192...), and
    `f4 == 0.1` for float32(0.1).  The device predicate rounds the constant to float32 first (round-2 ADVICE: comparing the widened
    column with the double constant dropped / added exactly the rows that sit on the constant — decimal data stored as float32)."""
    rng = np.random.default_rng(8)
    n = 400_000
    pool = np.array([0.1, 0.3, 0.30000001, 0.29999998, 0.5, 0.7, np.nan, -0.1], dtype="f4")
    cols = dict(x=rng.normal(0, 1, n), y=rng.normal(0, 1, n), f=rng.choice(pool, n), g=rng.choice(pool, n).astype("f8"))
    f = Frame(cols, chunk_size=100_000, nthreads=2)
    with np.errstate(invalid="ignore"):
        for expr, keep in (("f <= 0.3", cols["f"] <= 0.3), ("f == 0.1", cols["f"] == 0.1), ("f > 0.1", cols["f"] > 0.1), ("f != 0.3", cols["f"] != 0.3),
                           ("f < 0.3", cols["f"] < 0.3), ("(f >= 0.3) & (g >= 0.3)", (cols["f"] >= 0.3) & (cols["g"] >= 0.3))):
            got = f.count(binby=["x", "y"], limits=LIM, shape=16, selection=expr, edges=True)
            assert int(got.sum()) == int(keep.sum()), (expr, int(got.sum()), int(keep.sum()))
            assert np.array_equal(_want_mask(expr, cols), keep), expr
    # the rows ON the constant are what the rounding decides
    assert int((cols["f"] == np.float32(0.3)).sum()) > 10_000 and int((cols["f"].astype("f8") <= 0.3).sum()) != int((cols["f"] <= 0.3).sum())


# ------------------------------------------------------------------------------------------------------------
# round 4: the selection FUSED into the binning kernels (BinArgs::pred) — one selection shared by every aggregator, its terms over
# one float64 column: no sel_eval pass, no mask byte.  Each shape is binned three ways — fused, through sel_eval's mask
# ("fuse_selection" = 0) and with a numpy-built mask: the three must agree bit for bit on integer grids (the same rows are kept).
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", ["three_d_128", "bench_2d_box", "count_2d_lds", "count_2d_box", "two_columns_count_box", "one_d_big", "two_terms_nan", "std_box", "not_fusable_int", "two_columns_3d", "two_columns_or_nan_3d", "two_columns_box", "two_columns_count_lds", "two_columns_1d", "not_fusable_three_columns"])
def test_selection_fused_into_the_binning_kernels(shape):
    import torch
    g = torch.Generator(device="cuda").manual_seed(77)
    n = (1 << 25) + 4_321
    x, y, z = (torch.randn(n, dtype=torch.float64, device="cuda", generator=g) for _ in range(3))
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    v[::1000] = float("nan")
    j = torch.randint(-100, 100, (n,), dtype=torch.int32, device="cuda", generator=g)
    f = Frame(dict(x=x, y=y, z=z, v=v, j=j))
    spec = {
        "three_d_128": ("count", None, ["x", "y", "z"], [[-4, 4]] * 3, 128, "v > 3", "part_scatter_wv", True),
        "bench_2d_box": ("mean", "v", ["x", "y"], LIM, 256, "v > 3", "part_scatter_", True),               # the selection's column IS the value column
        "count_2d_lds": ("count", None, ["x", "y"], LIM, 256, "z <= 0.5", "count_lds", True),             # (count_box_pct = 0: the packed-counter LDS kernel)
        "count_2d_box": ("count", None, ["x", "y"], LIM, 256, "z <= 0.5", "part_scatter_phased_hot", True),   # round 5: the same call's default road, through the hot box
        "two_columns_count_box": ("count", None, ["x", "y"], LIM, 256, "(z <= 0.5) | (v > 4)", "part_scatter_phased_hot", True),
        "one_d_big": ("sum", "v", ["x"], [[-4, 4]], 100_000, "y != 0.25", "part_scatter", True),
        "two_terms_nan": ("count", None, ["x", "y", "z"], [[-4, 4]] * 3, 128, "~(v >= 1) | (v == 3.5)", "part_scatter_wv", True),   # NaN rows: kept by ~(v >= 1)
        "std_box": ("std", "v", ["x", "y"], LIM, 256, "(z > -1) & (z < 2)", "part_scatter_", True),
        "not_fusable_int": ("count", None, ["x", "y", "z"], [[-4, 4]] * 3, 128, "j > 3", "part_scatter_wv", False),
        # round 5: terms over TWO float64 columns are evaluated in the kernels too (PredDesc::col2; MASKED = 4)
        "two_columns_3d": ("count", None, ["x", "y", "z"], [[-4, 4]] * 3, 128, "(v > 3) & (x < 1)", "part_scatter_wv", True),
        "two_columns_or_nan_3d": ("count", None, ["x", "y", "z"], [[-4, 4]] * 3, 128, "~(v >= 1) | (y > 0.5)", "part_scatter_wv", True),
        "two_columns_box": ("mean", "v", ["x", "y"], LIM, 256, "(z > -1) & (v < 5)", "part_scatter_phased_hot", True),   # next to the hot box (the phased grouped form)
        "two_columns_count_lds": ("count", None, ["x", "y"], LIM, 256, "(z <= 0.5) | (v > 4)", "count_lds", True),
        "two_columns_1d": ("sum", "v", ["x"], [[-4, 4]], 100_000, "(y != 0.25) & (z < 1)", "part_scatter", True),
        "not_fusable_three_columns": ("count", None, ["x", "y", "z"], [[-4, 4]] * 3, 128, "(v > 3) & (x < 1) & (y > -1)", "part_scatter_wv", False),
    }[shape]
    what, col, binby, lim, shp, expr, kernel_prefix, fusable = spec
    call = lambda sel: (getattr(f, what)(binby=binby, limits=lim, shape=shp, selection=sel, edges=True) if col is None else getattr(f, what)(col, binby=binby, limits=lim, shape=shp, selection=sel, edges=True))
    f0, m0 = sa.config_get("pred_fused"), sa.config_get("pred_materialized")
    if len(binby) == 3:
        sa.config_set("strategy", 4)   # (33 M rows into 2.2 M cells: the planner would take device atomics; BASELINE configs[2]'s 1e9 rows take the partition)
    if shape.endswith("count_lds") or shape == "count_2d_lds":
        sa.config_set("count_box_pct", 0)   # (keep the call on count_lds_f64: its fused instantiations are what these two shapes test)
    try:
        fused = np.asarray(call(expr))
        kernel = sa.last_kernel(0)
        df, dm = sa.config_get("pred_fused") - f0, sa.config_get("pred_materialized") - m0
        assert kernel.startswith(kernel_prefix), kernel
        assert (df > 0 and dm == 0) if fusable else (df == 0), (shape, df, dm, kernel)
        sa.config_set("fuse_selection", 0)
        try:
            through_mask = np.asarray(call(expr))
        finally:
            sa.config_set("fuse_selection", 1)
        cols = dict(x=x, y=y, z=z, v=v, j=j)
        keep = _want_mask(expr, {k: t.cpu().numpy() for k, t in cols.items() if k in expr})
        numpy_mask = np.asarray(call(torch.from_numpy(keep.astype(np.uint8)).cuda()))
    finally:
        sa.config_set("strategy", 0)
        sa.config_set("count_box_pct", 90)
    _compare_three(shape, what, fused, through_mask, numpy_mask, keep)


def _compare_three(shape, what, fused, through_mask, numpy_mask, keep):
    if fused.dtype.kind in "iu":
        assert np.array_equal(fused, through_mask) and np.array_equal(fused, numpy_mask), shape
        assert int(fused.sum()) == int(keep.sum())
    else:   # mean / sum / std of the same kept rows: the kernels differ in their order of addition only
        for other in (through_mask, numpy_mask):
            if what == "std":   # (a cell with one kept row: s2/n - mean^2 is rounding noise around zero, its root NaN or 1e-8)
                a, b = np.nan_to_num(fused, nan=0.0), np.nan_to_num(other, nan=0.0)
                assert np.allclose(a, b, rtol=1e-6, atol=1e-5), (shape, float(np.max(np.abs(a - b))))
            else:
                assert np.array_equal(np.isnan(fused), np.isnan(other)), shape
                assert np.allclose(fused, other, rtol=1e-11, atol=2e-10, equal_nan=True), (shape, float(np.nanmax(np.abs(fused - other))))


# ------------------------------------------------------------------------------------------------------------
# round 5: arithmetic over float64 columns and virtual columns — the term's left side is a postfix program evaluated per row on the
# device (vxh_selection_set_program: + - * / negate, square, sqrt, abs).  IEEE makes each of these correctly rounded, so the device
# must keep EXACTLY the rows numpy keeps; the columns below carry the values where that is decided (zeros of both signs, infinities,
# NaN, subnormals, 1 ulp around the constants, negative arguments of sqrt, division by zero).
# ------------------------------------------------------------------------------------------------------------
ARITH_EXPRS = [
    ("x**2 + y**2 < 2", None), ("2*x + 1 > 0", None), ("(rs < 1.5) & (v > 2)", {"rs": "sqrt(x**2 + y**2)"}), ("abs(x - y) / 3 >= 0.25", None),
    ("-x <= 0", None), ("x / y > 2", None), ("1 / x < -3", None), ("sqrt(x) >= 0.7071067811865476", None), ("sqrt(e) == 1.4142135623730951", None),
    ("(e * e - 2) / 3 != 0", None), ("x * y - v > -2.5", None), ("e - 1.4142135623730951 > 0", None), ("((x + y) * (x - y)) / (1 + x**2) <= 0.1", None),
    ("(r2 < 1) | ~(abs(v - 3) > 0.5)", {"r2": "x*x + y*y"}),
]


@pytest.mark.parametrize("where", ["host", "device"])
def test_arithmetic_and_virtual_column_selections_keep_the_rows_numpy_keeps(where):
    import torch
    n = 2_000_003
    cols = {k: c for k, c in _columns(n, 4).items() if k in ("x", "y", "v")}
    rng = np.random.default_rng(9)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, 1.0, -1.0, 0.5, 2.0, 1e308, -1e308, 1e-200, 1e200])
    for k in ("x", "y"):
        cols[k][rng.integers(0, n, 20_000)] = rng.choice(special, 20_000)
    e = np.full(n, np.sqrt(2.0))
    e[::3] = np.nextafter(np.sqrt(2.0), 2.0); e[1::3] = np.nextafter(np.sqrt(2.0), 0.0)
    e[::7] = 2.0; e[::11] = np.nextafter(2.0, 3.0); e[::13] = np.nextafter(2.0, 1.0)
    cols["e"] = e
    held = {k: (torch.from_numpy(c).cuda() if where == "device" else c) for k, c in cols.items()}
    f = Frame(held, chunk_size=1 << 19, nthreads=3)
    ns = dict(cols, sqrt=np.sqrt, abs=np.abs)
    for expr, virtual in ARITH_EXPRS:
        pred = P.compile_selection(expr, cols, virtual=virtual)
        assert pred.programs, expr
        env = dict(ns)
        for name, ve in (virtual or {}).items():
            env[name] = eval(ve, {}, env)
        with np.errstate(all="ignore"):
            keep = np.asarray(eval(expr, {}, env), dtype=bool)
        assert np.array_equal(pred.numpy_mask(cols), keep), expr
        f._predicates[expr] = pred                      # (Frame compiles a string itself; a virtual column's expression comes with the Predicate)
        got = f.count(binby=["v"], limits=[[-5, 11]], shape=64, selection=expr, edges=True)
        mask = torch.from_numpy(keep.astype(np.uint8)).cuda() if where == "device" else keep
        want = f.count(binby=["v"], limits=[[-5, 11]], shape=64, selection=mask, edges=True)
        assert np.array_equal(got, want), (expr, int(np.abs(np.asarray(got) - np.asarray(want)).sum()))
        assert int(np.asarray(got).sum()) == int(keep.sum()), expr


def test_arithmetic_selections_through_the_ready_made_mask_passes_on_device_columns():
    """the passes that take a ready-made keep-mask — minmax, the percentile / limits passes, the fused hash groupby —
    built it for DEVICE columns from (first column, op, constant) of every term and dropped the term's arithmetic program: `2*x + 1 > 0`
    became `x > 0`.  Device columns against the same calls over host columns and against numpy."""
    import torch
    from vaex_amd.binned import agg
    n = 1_500_007
    rng = np.random.default_rng(21)
    x, y, v = rng.normal(0, 1, n), rng.normal(0, 1, n), rng.normal(3, 2, n)
    x[::701] = np.nan
    k = (rng.integers(0, 5000, n) * 2654435761) % (1 << 40)      # scattered keys: the fused hash pass
    kd = rng.integers(0, 300, n)                                 # a dense range: the ordinal pass
    host = dict(x=x, y=y, v=v, k=k, kd=kd)
    dev = {c: torch.from_numpy(a).cuda() for c, a in host.items()}
    fh, fd = Frame(host, chunk_size=1 << 19, nthreads=3), Frame(dev)
    for expr in ("2*x + 1 > 0", "x**2 + y**2 < 4", "(abs(x - y)/2 >= 0.25) & (v > 3)", "sqrt(x*x + 1) - y < 1.5"):
        pred = P.compile_selection(expr, host)
        assert pred.programs, expr
        with np.errstate(all="ignore"):
            keep = np.asarray(eval(expr, {}, dict(host, sqrt=np.sqrt, abs=np.abs)), dtype=bool)
        plain = P.Predicate(expr, pred.columns, pred.terms, pred.truth).numpy_mask(host)   # (the programs dropped: what the bug computed)
        assert not np.array_equal(plain, keep), expr
        assert np.array_equal(fd._mask_array(expr).cpu().numpy().astype(bool), keep), expr
        # minmax
        want = np.array([np.nanmin(v[keep]), np.nanmax(v[keep])])
        assert np.array_equal(fd.minmax("v", selection=expr), want) and np.array_equal(fh.minmax("v", selection=expr), want), expr
        # groupby: scattered keys (gb_scatter + gb_reduce with the mask as keep), and a dense range
        for key in ("k", "kd"):
            spec = {"c": agg.count("v"), "s": agg.sum("v"), "n": agg.count()}
            rd, rh = fd.groupby(key, spec, selection=expr), fh.groupby(key, spec, selection=expr)
            uniq, inv = np.unique(host[key][keep], return_inverse=True)
            assert np.array_equal(rd[key], uniq) and np.array_equal(rh[key], uniq), (expr, key)
            assert np.array_equal(rd["n"], np.bincount(inv)) and np.array_equal(rd["c"], rh["c"]), (expr, key)
            ssum = np.bincount(inv, weights=v[keep])
            assert np.allclose(rd["s"], ssum, rtol=1e-12, atol=1e-9) and np.allclose(rh["s"], ssum, rtol=1e-12, atol=1e-9), (expr, key)
        # the aggregations' OWN selection (groups of all rows, values of the kept rows): the fused pass run twice
        own = fd.groupby("k", {"c": agg.count(selection=expr), "s": agg.sum("v", selection=expr)})
        uniq_all, inv_all = np.unique(k, return_inverse=True)
        assert np.array_equal(own["k"], uniq_all) and np.array_equal(own["c"], np.bincount(inv_all, weights=keep, minlength=len(uniq_all)).astype(np.int64)), expr


def test_random_predicates_keep_the_rows_numpy_keeps():
    """the differential fuzz of tests/test_predicate.py on the device: random expressions (comparisons of every column type with integer / float
    / huge / float32-boundary constants on either side, arithmetic over float64 columns, & | ~ three levels deep) as device predicates — in the
    kernels where the call's shape allows, through sel_eval otherwise — against the same call with numpy's keep-mask"""
    from tests.predicate_fuzz import random_expression
    cols = _columns(150_001, 9)
    cols["x"][5], cols["x"][6], cols["x"][7] = np.inf, -np.inf, -0.0
    cols["f"][::17] = np.nan
    names = [k for k in cols if k != "t"] + ["t"]
    f = Frame(cols, chunk_size=40_000, nthreads=2)
    inside = 0
    for seed in range(260):
        expr = random_expression(np.random.default_rng(1000 + seed), names, ["x", "y", "v"])
        try:
            keep = _want_mask(expr, cols)
        except P.Unsupported:
            continue
        inside += 1
        shape = 64 if seed % 3 else 300                      # (LDS-resident count kernels / the partition pass)
        got = f.count(binby=["y", "v"], limits=[[-4, 4], [-3, 9]], shape=shape, selection=expr, edges=True)
        want = f.count(binby=["y", "v"], limits=[[-4, 4], [-3, 9]], shape=shape, selection=keep, edges=True)
        assert np.array_equal(got, want), expr
        assert int(got.sum()) == int(keep.sum()), expr
    assert inside > 150, inside


KNOBS = ("strategy", "wv", "hot_min_rows", "hot_min_pct", "convert_binners", "count_box_pct", "part_chunk")


@pytest.mark.parametrize("block", range(int(__import__("os").environ.get("VAEX_AMD_FUZZ_BLOCKS", "6"))))   # (14 calls a block; a soak run raises it)
def test_random_calls_with_predicates_over_every_column_kind_and_kernel_form(block):
    """the same differential fuzz over what round 5 added to the kernels: 1-3 binner columns of any kind (float64 / float32 / int32 / int16 /
    uint8 — converted on load or by the pre-pass), count / sum / mean / min of a float64 or float32 column, up to 1.2e6 rows with the
    thresholds pulled down so that the partition passes, the hot box (grouped, phased and ring-less forms) and the conversion pre-pass all run —
    each call once with its random selection as a device predicate (inside the binning kernels where the shape allows) and once with numpy's
    keep-mask of the same expression"""
    import vaex_amd
    from tests.predicate_fuzz import random_expression
    sa = vaex_amd.superagg
    defaults = {k: sa.config_get(k) for k in KNOBS}
    limits = dict(x=[-4, 4], y=[-4, 4], v=[-3, 9], f=[-3, 3], j=[-100, 100], h=[-100, 100], c=[0, 256])
    frames = {}
    try:
        for seed in range(block * 14, block * 14 + 14):
            rng = np.random.default_rng(7000 + seed)
            n = int(rng.choice([5_000, 150_001, 1_200_003]))
            if n not in frames:
                cols = _columns(n, 11)
                cols["v"][::333] = np.nan
                frames[n] = (cols, Frame(cols, chunk_size=1 << 22, nthreads=2))
            cols, f = frames[n]
            nd = int(rng.choice([1, 2, 2, 3]))
            binby = [str(b) for b in rng.choice(list(limits), size=nd, replace=False)]
            shape = int(rng.choice([16, 64, 300, 700])) if nd < 3 else int(rng.choice([16, 64, 128]))
            expr = None
            for t in range(30):
                e = random_expression(np.random.default_rng(int(rng.integers(1 << 30))), [k for k in cols if k != "U"], ["x", "y", "v"])
                try:
                    keep = _want_mask(e, cols)
                    expr = e
                    break
                except P.Unsupported:
                    continue
            assert expr is not None
            sa.config_set("strategy", int(rng.choice([0, 0, 4])))
            sa.config_set("wv", int(rng.choice([6, 6, 5, 3, 0])))
            if rng.random() < 0.6:
                sa.config_set("hot_min_rows", 1)
                sa.config_set("hot_min_pct", 5)
            if rng.random() < 0.5:
                sa.config_set("convert_binners", 1000)
            if rng.random() < 0.3:
                sa.config_set("count_box_pct", 0)
            if rng.random() < 0.3:
                sa.config_set("part_chunk", 1 << 20)
            stat = str(rng.choice(["count", "count", "sum", "mean", "min", "sumf"]))
            kw = dict(binby=binby, limits=[limits[b] for b in binby], shape=shape, edges=True)
            if stat == "count":
                got, want = f.count(selection=expr, **kw), f.count(selection=keep, **kw)
            elif stat == "sumf":
                got, want = f.sum("f", selection=expr, **kw), f.sum("f", selection=keep, **kw)
            else:
                got, want = getattr(f, stat)("v", selection=expr, **kw), getattr(f, stat)("v", selection=keep, **kw)
            what = (seed, n, binby, shape, stat, expr, {k: sa.config_get(k) for k in KNOBS})
            if stat in ("count", "min"):
                assert np.array_equal(got, want, equal_nan=True), what
            else:
                fin = np.abs(want[np.isfinite(want)])
                assert np.allclose(got, want, rtol=1e-12, atol=1e-12 * max(1.0, float(fin.max()) if fin.size else 1.0), equal_nan=True), what
            if stat == "count":
                assert int(got.sum()) == int(keep.sum()), what
            for k, val in defaults.items():
                sa.config_set(k, val)
    finally:
        for k, val in defaults.items():
            sa.config_set(k, val)


@pytest.mark.parametrize("ncols", [1, 2, 3, 4])
def test_the_quad_form_of_sel_eval_keeps_the_rows_numpy_keeps(ncols):
    """round 6 (late): selections whose terms all compare columns with constants take sel_eval_vec (whole quads of rows, every load
    issued before the first comparison; one or two columns — more go through the generic kernel) — here float64 columns: the same bytes as numpy for every length around a quad, for columns that start off a 16-byte
    boundary (those go through the generic kernel), for NaN / signed zeros / infinities, with every comparison and truth table"""
    import torch
    import vaex_amd
    sa = vaex_amd.superagg
    rng = np.random.default_rng(40 + ncols)
    ops = [np.less, np.less_equal, np.greater, np.greater_equal, np.equal, np.not_equal]
    for n in (1, 3, 4, 5, 7, 8, 1023, 1024, 100_003, 3_000_001):
        for offset in (0, 1):   # (1: the columns start 8 bytes off a 16-byte boundary)
            host = []
            for c in range(ncols):
                a = rng.normal(0, 1, n + offset)
                a[rng.random(n + offset) < 0.05] = np.nan
                a[rng.random(n + offset) < 0.05] = 0.0
                a[rng.random(n + offset) < 0.02] = -0.0
                a[rng.random(n + offset) < 0.01] = np.inf
                a[rng.random(n + offset) < 0.01] = -np.inf
                host.append(a)
            dev = [torch.from_numpy(a).cuda()[offset:] for a in host]
            host = [a[offset:] for a in host]
            for trial in range(6):
                nterms = int(rng.integers(max(1, ncols), 5))
                cols_of = list(range(ncols)) + [int(rng.integers(0, ncols)) for _ in range(nterms - ncols)]   # every column is read by a term
                terms = [(cols_of[t], int(rng.integers(0, 6)), float(rng.choice([0.0, -0.0, 0.5, -1.0, np.inf, -np.inf, np.nan, 1e-300]))) for t in range(nterms)]
                truth = int(rng.integers(0, 1 << (1 << nterms)))
                sel = sa.Selection(1, [0] * ncols, terms, truth)
                for c in range(ncols):
                    sel.set_data(0, c, dev[c])
                out = torch.full(((n + 3) & ~3,), 7, dtype=torch.uint8, device="cuda")
                sel.evaluate(0, n, out)
                sa.slot_wait(0)
                bits = np.zeros(n, dtype=np.uint32)
                with np.errstate(invalid="ignore"):
                    for t, (c, op, value) in enumerate(terms):
                        bits |= ops[op](host[c], np.float64(value)).astype(np.uint32) << t
                want = ((truth >> bits) & 1).astype(np.uint8)
                np.testing.assert_array_equal(out[:n].cpu().numpy(), want, err_msg=f"n={n} offset={offset} terms={terms} truth={truth}")


def test_the_quad_form_of_sel_eval_is_the_generic_kernel_for_every_dtype():
    """sel_eval_vec compares from registers by the generic kernel's own rule (term_cmp_bits): the same selection over the same values held
    ONE ELEMENT off the quad form's alignment goes through the generic kernel — the two masks are the same bytes, for every column dtype,
    integer and float constants, one and two columns, with and without the missing-value mask behind it"""
    import torch
    import vaex_amd
    sa = vaex_amd.superagg
    rng = np.random.default_rng(77)
    kinds = [("float64", 0), ("float32", 1), ("int64", 2), ("int32", 3), ("int16", 4), ("int8", 5), ("uint64", 6), ("uint32", 7), ("uint16", 8), ("uint8", 9), ("bool", 10)]

    def column(kind, n):
        if kind.startswith("float"):
            a = rng.normal(0, 2, n).astype(kind)
            a[rng.random(n) < 0.05] = np.nan
            a[rng.random(n) < 0.05] = 0.0
            return a
        if kind == "bool":
            return rng.random(n) < 0.5
        info = np.iinfo(kind)
        a = rng.integers(max(info.min, -4), min(info.max, 4) + 1, n).astype(kind)
        a[rng.random(n) < 0.02] = info.max
        a[rng.random(n) < 0.02] = info.min
        return a
    checked = 0
    for trial in range(120):
        ncols = 1 + trial % 2
        n = int(rng.choice([4, 5, 63, 1024, 100_003, 1_000_001]))
        picked = [kinds[int(rng.integers(0, len(kinds)))] for _ in range(ncols)]
        host = [column(k, n + 1) for k, _ in picked]
        on_quads = [torch.from_numpy(a[1:].copy()).cuda() for a in host]       # (fresh allocations: aligned)
        off_quads = [torch.from_numpy(a).cuda()[1:] for a in host]              # (the same values one element further)
        nterms = int(rng.integers(ncols, 5))
        cols_of = list(range(ncols)) + [int(rng.integers(0, ncols)) for _ in range(nterms - ncols)]
        consts = [0, 1, -1, 2, 3, -4, 127, -128, 255, 65535, 2 ** 31 - 1, -2 ** 31, 2 ** 63 - 1, -2 ** 63, 0.0, 0.5, -1.5, 2.0, float("nan"), float("inf"), 1e-3]
        terms = [(cols_of[t], int(rng.integers(0, 6)), consts[int(rng.integers(0, len(consts)))]) for t in range(nterms)]
        truth = int(rng.integers(0, 1 << (1 << nterms)))
        outs = []
        for cols in (on_quads, off_quads):
            sel = sa.Selection(1, [code for _, code in picked], terms, truth)
            for c in range(ncols):
                sel.set_data(0, c, cols[c])
            out = torch.full(((n + 3) & ~3,), 9, dtype=torch.uint8, device="cuda")
            sel.evaluate(0, n, out)
            sa.slot_wait(0)
            outs.append(out[:n].cpu().numpy())
        np.testing.assert_array_equal(outs[0], outs[1], err_msg=f"n={n} dtypes={[k for k, _ in picked]} terms={terms} truth={truth}")
        assert set(np.unique(outs[0])) <= {0, 1}
        checked += 1
    assert checked == 120
