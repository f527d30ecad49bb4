"""-m gpu: parity of the HIP path (through the C-ABI, via the superagg-compatible shim) against the
oracle: the C restatement always, and the reference's own compiled C++ (oracle/_ref) when present.
Integer grids bit-exact; fp64 sums within 1e-12 of sum|v| per cell (BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import oracle
from tests import cases

pytestmark = pytest.mark.gpu

STRATEGIES = {"auto": 0, "global": 1, "xcc": 2, "lds": 3, "part": 4}


@pytest.fixture(autouse=True)
def _reset_config(sa, gpu_ready):
    def reset():
        sa.config_set("strategy", 0)
        sa.config_set("block", 0)
        sa.config_set("blocks", 0)
        sa.config_set("slab_log2", -1)
        sa.config_set("part_chunk", 0)
        sa.config_set("parts", 0)
        sa.config_set("count16", 1)
    reset()
    yield
    reset()


def check(sa, case, **kw):
    want = oracle.run_case(case)
    got = cases.run_superagg(sa, case, **kw)
    cases.assert_case_equal(got, want, case)
    return got


def test_count_1d_golden(sa):
    # reference tests/agg_test.py:150-158
    x = np.array([-1, -2, 0.5, 1.5, 4.5, 5], dtype="f8")
    case = dict(n=6, binners=[dict(kind="scalar", data=x, vmin=0, vmax=5, bins=5)], aggs=[dict(kind="count")])
    got = cases.run_superagg(sa, case)
    assert got[0].tolist() == [0, 2, 1, 1, 0, 0, 1, 1]


def test_count_1d_ordinal_golden(sa):
    # reference tests/agg_test.py:171-180
    x = np.array([-1, -2, 0, 1, 4, 5], dtype="i8")
    case = dict(n=6, binners=[dict(kind="ordinal", data=x, count=5)], aggs=[dict(kind="count")])
    got = cases.run_superagg(sa, case)
    assert got[0].tolist() == [1, 1, 0, 0, 1, 3, 0]


@pytest.mark.parametrize("strategy", ["auto", "global", "xcc", "lds", "part"])
@pytest.mark.parametrize("n", [1, 63, 1000, 200_003])
def test_2d_count_mean(sa, strategy, n):
    sa.config_set("strategy", STRATEGIES[strategy])
    case = cases.case_2d_count_mean(n, shape=32)
    check(sa, case)
    assert sa.last_kernel(0) != ""


@pytest.mark.parametrize("strategy", ["global", "xcc", "lds", "part"])
def test_2d_256_count_mean_selection(sa, strategy):
    sa.config_set("strategy", STRATEGIES[strategy])
    case = cases.case_2d_count_mean(300_000, shape=256, selection=True)
    check(sa, case)


@pytest.mark.parametrize("strategy", ["auto", "part"])
def test_3d_selection(sa, strategy):
    sa.config_set("strategy", STRATEGIES[strategy])
    case = cases.case_3d_selection(250_000, shape=32)
    check(sa, case)


def test_part_many_slabs_3d(sa):
    # 67^3 = 300k cells of u32 = 1.2 MB -> 8+ slabs; records are just a uint16 local index
    sa.config_set("strategy", STRATEGIES["part"])
    case = cases.case_3d_selection(2_000_000, shape=64)
    check(sa, case)
    assert sa.last_kernel(0).startswith("part_scatter")


def test_part_mixed_masks_and_inputs(sa):
    # aggregators with different masks / inputs -> records carry a flags byte and several value columns
    sa.config_set("strategy", STRATEGIES["part"])
    c = cases.gaussian_columns(400_000, seed=5)
    m1 = c["v"] > 3
    m2 = c["z"] < 0
    case = dict(n=400_000, binners=[dict(kind="scalar", data=c["x"], vmin=-4, vmax=4, bins=300), dict(kind="scalar", data=c["y"], vmin=-4, vmax=4, bins=300)],
                aggs=[dict(kind="count"), dict(kind="count", mask=m1), dict(kind="sum", data=c["v"], mask=m2), dict(kind="sum", data=c["z"], mask=m1),
                      dict(kind="max", data=c["v"]), dict(kind="min", data=c["z"].astype("f4"), mask=m2), dict(kind="summoment", data=c["v"], moment=2)])
    check(sa, case)


def test_part_chunks_and_queue_overflow(sa):
    # small chunks + every row in ONE cell: the slab queue overflows and the HBM-atomic slow path takes over
    sa.config_set("strategy", STRATEGIES["part"])
    sa.config_set("part_chunk", 1 << 20)
    n = 3_000_000
    x = np.full(n, 0.5); y = np.full(n, -0.25); v = np.arange(n, dtype="f8") % 7
    case = dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=y, vmin=-4, vmax=4, bins=256)], aggs=[dict(kind="count"), dict(kind="sum", data=v)])
    got = check(sa, case)
    assert got[0].max() == n
    case2 = cases.case_2d_count_mean(2_500_000, shape=256)
    check(sa, case2)
    check(sa, case2, chunk=700_000, nthreads=3)


def test_part_groupby_100k(sa):
    sa.config_set("strategy", STRATEGIES["part"])
    case = cases.case_groupby(3_000_000, groups=100_000)
    check(sa, case)


@pytest.mark.parametrize("strategy", ["auto", "global", "lds", "part"])
def test_groupby_ordinal(sa, strategy):
    sa.config_set("strategy", STRATEGIES[strategy])
    case = cases.case_groupby(200_000, groups=1000)
    check(sa, case)


def test_empty_and_tiny(sa):
    case = cases.case_2d_count_mean(5, shape=4)
    case["n"] = 0
    got = cases.run_superagg(sa, case)
    assert got[0].sum() == 0 and got[1].sum() == 0
    case["n"] = 5
    check(sa, case)


@pytest.mark.parametrize("chunk,nthreads", [(1000, 1), (4096, 3), (77, 2)])
def test_chunks_and_slots(sa, chunk, nthreads):
    case = cases.case_2d_count_mean(20_000, shape=16)
    check(sa, case, chunk=chunk, nthreads=nthreads)


def test_device_resident_columns(sa):
    case = cases.case_2d_count_mean(100_000, shape=64, selection=True)
    check(sa, case, to_device=cases.torch_device_array)
    check(sa, case, to_device=cases.torch_device_array, chunk=30_000, nthreads=2)


ALL_DTYPES = ["float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"]


def _typed(rng, n, dtype, lo=-20, hi=20):
    if dtype == "bool":
        return rng.integers(0, 2, n).astype(bool)
    if dtype.startswith("float"):
        a = rng.normal(0, 8, n).astype(dtype)
        a[rng.integers(0, n, 5)] = np.nan
        return a
    if dtype.startswith("uint"):
        return rng.integers(0, hi, n).astype(dtype)
    return rng.integers(lo, hi, n).astype(dtype)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("flip", [False, True])
def test_all_dtypes_scalar_binner_and_aggs(sa, dtype, flip):
    rng = np.random.default_rng(7)
    n = 5000
    x = _typed(rng, n, dtype)
    v = _typed(rng, n, dtype)
    if flip:
        if np.dtype(dtype).itemsize == 1:
            pytest.skip("single-byte types have no byte order")
        x = x.astype(np.dtype(dtype).newbyteorder(">"))
        v = v.astype(np.dtype(dtype).newbyteorder(">"))
    bmask = rng.random(n) < 0.1
    amask = rng.random(n) < 0.7
    case = dict(n=n, binners=[dict(kind="scalar", data=x, mask=bmask, vmin=-10, vmax=10, bins=7)],
                aggs=[dict(kind="count", data=v), dict(kind="count", data=v, mask=amask), dict(kind="sum", data=v, mask=amask), dict(kind="sum", data=v),
                      dict(kind="summoment", data=v, moment=2), dict(kind="min", data=v, mask=amask), dict(kind="max", data=v)])
    check(sa, case)


@pytest.mark.parametrize("dtype", ALL_DTYPES)
@pytest.mark.parametrize("allow_other,invert", [(False, False), (True, False), (False, True), (True, True)])
def test_ordinal_binner_all_dtypes(sa, dtype, allow_other, invert):
    rng = np.random.default_rng(11)
    n = 4000
    k = _typed(rng, n, dtype, lo=-3, hi=12)
    v = rng.normal(0, 1, n)
    bmask = rng.random(n) < 0.05
    case = dict(n=n, binners=[dict(kind="ordinal", data=k, mask=bmask, count=8, min_value=1, allow_other=allow_other, invert=invert)], aggs=[dict(kind="count"), dict(kind="sum", data=v)])
    check(sa, case)


def test_mixed_dims_scalar_ordinal(sa):
    rng = np.random.default_rng(3)
    n = 50_000
    x = rng.normal(0, 1, n).astype("f4")
    k = rng.integers(0, 5, n).astype("i2")
    z = rng.integers(-5, 5, n).astype("i8")
    v = rng.normal(0, 1, n)
    case = dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-2, vmax=2, bins=10), dict(kind="ordinal", data=k, count=5), dict(kind="scalar", data=z, vmin=-5, vmax=5, bins=5)],
                aggs=[dict(kind="count"), dict(kind="sum", data=v), dict(kind="max", data=x)])
    check(sa, case)


def test_against_reference_cpp(sa, ref):
    """HIP vs the reference's own C++ (same call sequence, both modules)."""
    for case in (cases.case_2d_count_mean(100_000, shape=64, selection=True), cases.case_groupby(50_000, groups=300), cases.case_3d_selection(50_000, shape=16)):
        want = cases.run_superagg(ref, case, chunk=10_000, nthreads=2)
        got = cases.run_superagg(sa, case, chunk=10_000, nthreads=2)
        cases.assert_case_equal(got, want, case)


def test_buffer_seed_merge_reset(sa):
    case = cases.case_2d_count_mean(10_000, shape=8)
    keep = []
    got = cases.run_superagg(sa, case, keep=keep, grids=2, nthreads=1)
    count = keep[0]
    buf = np.asarray(count)  # (grids, 11, 11), dim-0-fastest strides
    assert buf.shape == (2, 11, 11)
    assert buf[0].T.flags.c_contiguous or buf[0].flags.f_contiguous
    np.testing.assert_array_equal(buf[0], got[0])
    assert buf[1].sum() == 0
    # seeding initial values through the buffer (vaex/cpu.py:658) is picked up by the next bin
    buf[0] += 5
    keep2 = []
    again = cases.run_superagg(sa, case, keep=keep2)
    count.merge([keep2[0]])
    np.testing.assert_array_equal(count.get_result(), 2 * got[0] + 5)
    count.reset()
    assert count.get_result().sum() == 0


def test_full_size_properties(sa):
    """BASELINE config sizes cannot be checked row by row against a CPU oracle in seconds; check the
    size-independent invariants on 2e8 device-generated rows: conservation of count, linearity of the
    sum under a split of the rows, and agreement of the three kernel strategies."""
    import torch
    n = 200_000_000
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    torch.cuda.synchronize()
    results = {}
    for strat in ("global", "part", "lds"):
        sa.config_set("strategy", STRATEGIES[strat])
        bx = sa.BinnerScalar_float64(1, "x", -4, 4, 256)
        by = sa.BinnerScalar_float64(1, "y", -4, 4, 256)
        grid = sa.Grid([bx, by])
        c = sa.AggCount_int64(grid, 1, 1)
        s = sa.AggSum_float64(grid, 1, 1)
        half = n // 2
        for lo, hi in ((0, half), (half, n)):
            bx.set_data(0, x[lo:hi]); by.set_data(0, y[lo:hi]); s.set_data(0, v[lo:hi], 0)
            bx.clear_data_mask(0); by.clear_data_mask(0); c.clear_data_mask(0); s.clear_data_mask(0)
            grid.bin(0, [c, s], hi - lo)
        results[strat] = (c.get_result(), s.get_result())
    cg, sg = results["global"]
    cx, sx = results["part"]
    assert cg.sum() == n
    np.testing.assert_array_equal(cg, cx)
    np.testing.assert_array_equal(cg, results["lds"][0])
    assert np.max(np.abs(sg - results["lds"][1])) <= 1e-12 * float(v.abs().sum()) / 1000
    total = float(v.sum())
    assert abs(sg.sum() - total) <= 1e-9 * float(v.abs().sum())
    assert np.max(np.abs(sg - sx)) <= 1e-12 * float(v.abs().sum()) / 1000
    inside = int(((x >= -4) & (x < 4) & (y >= -4) & (y < 4)).sum())
    assert cg[2:-1, 2:-1].sum() == inside


def test_rccl_allreduce_path_single_rank(sa):
    """The N>1 bench path on one GPU: a 1-rank RCCL group, all-reduce forced.  Exercises exactly what the
    multi-GPU run does per rank — alias the library-owned device grid through __cuda_array_interface__,
    dist.all_reduce (sum / min / max) in place, device_touch, get_result — and must leave the results unchanged."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from vaex_amd import dist as vdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        case = cases.case_groupby(300_000, groups=500)
        case["aggs"].append(dict(kind="count"))
        keep = []
        before = cases.run_superagg(sa, case, keep=keep, to_device=cases.torch_device_array)
        aggs = keep[:-1]
        vdist.allreduce_aggs(aggs, force=True)
        after = [np.array(a.get_result()) for a in aggs]
        for b, a in zip(before, after):
            np.testing.assert_array_equal(a, b)
        case2 = cases.case_2d_count_mean(2_000_000, shape=256)
        keep2 = []
        before2 = cases.run_superagg(sa, case2, keep=keep2, to_device=cases.torch_device_array)
        vdist.allreduce_aggs(keep2[:-1], force=True)
        for b, a in zip(before2, keep2[:-1]):
            np.testing.assert_array_equal(np.array(a.get_result()), b)
        assert vdist.Comm().minmax(3, 9) == (3, 9)
    finally:
        vdist._NATIVE.clear()
        dist.destroy_process_group()


def test_vxh_allreduce_native_world1(sa, monkeypatch):
    """round 4: the C-ABI's own collective (vxh_comm_unique_id / vxh_comm_init / vxh_allreduce on RCCL directly) with one
    rank: replicas folded and all-reduced in place on the library's stream — count / sum / moment (ncclSum on int64 / fp64),
    min / max (ncclMin / ncclMax, float32 cells included), twice in a row, with binning in between (stream order, no host stop);
    results unchanged.  No torch.distributed anywhere."""
    comm = sa.Comm(1, 0, sa.comm_unique_id())
    assert (comm.size, comm.rank) == (1, 0)
    c = cases.gaussian_columns(400_000, seed=21)
    v32 = c["v"].astype("f4")
    case = dict(n=400_000, binners=[dict(kind="scalar", data=c["x"], vmin=-4, vmax=4, bins=64), dict(kind="scalar", data=c["y"], vmin=-4, vmax=4, bins=64)],
                aggs=[dict(kind="count"), dict(kind="sum", data=c["v"]), dict(kind="summoment", data=c["v"], moment=2), dict(kind="min", data=c["v"]),
                      dict(kind="max", data=v32), dict(kind="count", data=c["v"])])
    keep = []
    before = cases.run_superagg(sa, case, keep=keep, to_device=cases.torch_device_array)
    aggs = keep[:-1]
    comm.allreduce(aggs)
    comm.allreduce(aggs[:2])
    for b, a in zip(before, aggs):
        np.testing.assert_array_equal(np.array(a.get_result()), b)
    # the torch path (round 3) still answers the same when asked for
    import os
    import socket
    import torch
    import torch.distributed as dist
    from vaex_amd import dist as vdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        monkeypatch.setenv("VAEX_AMD_TORCH_ALLREDUCE", "1")
        assert vdist.native_comm() is None
        vdist.allreduce_aggs(aggs[:3], force=True)
        monkeypatch.delenv("VAEX_AMD_TORCH_ALLREDUCE")
        assert vdist.native_comm() is not None
        vdist.allreduce_aggs(aggs[:3], force=True)
        for b, a in zip(before[:3], aggs[:3]):
            np.testing.assert_array_equal(np.array(a.get_result()), b)
    finally:
        vdist._NATIVE.clear()
        dist.destroy_process_group()


# ---- packed 16-bit LDS counters (all-count passes whose grid fits LDS only as uint16 halves) -----------------
def test_count16_2d_256(sa):
    c = cases.gaussian_columns(1_500_000, seed=9)
    sel = c["v"] > 3
    nanv = c["v"].copy(); nanv[::7] = np.nan
    bin2 = [dict(kind="scalar", data=c["x"], vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=c["y"], vmin=-4, vmax=4, bins=256)]
    for aggs in ([dict(kind="count")], [dict(kind="count", mask=sel)], [dict(kind="count", data=nanv)], [dict(kind="count", data=nanv, mask=sel)]):
        check(sa, dict(n=1_500_000, binners=bin2, aggs=aggs))
        assert sa.last_kernel(0).startswith(("bin_lds", "count_lds")), sa.last_kernel(0)
    # non-float64 binner column: the generic (typed) instantiation of the same kernel
    bin2f = [dict(kind="scalar", data=c["x"].astype("f4"), vmin=-4, vmax=4, bins=256), bin2[1]]
    check(sa, dict(n=1_500_000, binners=bin2f, aggs=[dict(kind="count")]))
    assert sa.last_kernel(0).startswith(("bin_lds", "count_lds"))
    # switched off: same numbers through partition + reduce
    sa.config_set("count16", 0)
    try:
        check(sa, dict(n=1_500_000, binners=bin2, aggs=[dict(kind="count")]))
        assert sa.last_kernel(0).startswith("part_scatter")
    finally:
        sa.config_set("count16", 1)


@pytest.mark.parametrize("pattern", ["one_even", "one_odd", "pair", "pair_skewed"])
def test_count16_half_word_wraps(sa, pattern):
    # 8 workgroups, ~2.4e5 rows each into one or two cells that share an LDS word: every half wraps 3+ times, the
    # carries into (and back out of) the odd half included.  Counts must still be exact.
    sa.config_set("blocks", 8)
    n = 2_000_000
    bins = 70_000  # 70 003 cells: 280 KB as uint32, 140 KB packed
    width = 1.0 / bins
    i = np.arange(n)
    even, odd = 1000, 1001  # interior sub-indices; cells are +2 -> LDS word 501 holds cells 1002 (even) and 1003 (odd)
    if pattern == "one_even":
        sub = np.full(n, even)
    elif pattern == "one_odd":
        sub = np.full(n, odd)
    elif pattern == "pair":
        sub = np.where(i % 2 == 0, even, odd)
    else:
        sub = np.where(i % 5 == 0, even, odd)
    x = (sub + 0.5) * width
    x[::1000] = 0.123  # a few rows elsewhere
    case = dict(n=n, binners=[dict(kind="scalar", data=x, vmin=0.0, vmax=1.0, bins=bins)], aggs=[dict(kind="count")])
    got = check(sa, case)
    assert sa.last_kernel(0).startswith(("bin_lds", "count_lds"))
    assert got[0].sum() == n and got[0].max() > 8 * 65536


def test_count16_partition_reduce_opt_in(sa):
    sa.config_set("count16", 2)
    sa.config_set("strategy", STRATEGIES["part"])
    try:
        case = cases.case_3d_selection(2_000_000, shape=64)
        check(sa, case)
        assert sa.last_kernel(0).startswith("part_scatter")
        n = 3_000_000  # one hot cell: sub-queue overflow path + half-word wraps in pass 2
        x = np.full(n, 0.5); y = np.full(n, -0.25)
        check(sa, dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4, vmax=4, bins=512), dict(kind="scalar", data=y, vmin=-4, vmax=4, bins=512)], aggs=[dict(kind="count")]))
    finally:
        sa.config_set("count16", 1)


# ---- hot box: pass 1 of the partition strategy aggregates the densest rectangle of cells in LDS ---------------
WV_DEFAULT = 6  # pass 1 next to a hot box: part_scatter_wv, cold records in slab-sorted groups held in registers and written in chip-wide bursts (5: written as they come; 3: without rings, one record stream per (wave, slab))


@pytest.fixture(params=[1, 2, 3, 4, 5, 6], ids=["blk", "wv_rings", "direct", "shared", "grouped", "phased"])
def hot_pass1(request, sa):
    """The hot-box tests run with every pass-1 kernel that can sit next to a box: part_scatter_blk ("wv" 1),
    part_scatter_wv with rings (2), without rings and one record stream per (wave, slab) (3) / per (workgroup, slab) (4), with slab-sorted
    groups in one stream per wave (5: <= 8 slabs, otherwise it is 3), the same with the groups held in registers between chip-wide write
    bursts (6, round 5: the default)."""
    sa.config_set("wv", request.param)
    yield request.param
    sa.config_set("wv", WV_DEFAULT)
    sa.config_set("wv_block", 0)


def _hot_reset(sa):
    for k in ("hot_x0", "hot_y0", "hot_w", "hot_h"):
        sa.config_set(k, 0)
    sa.config_set("hot", 1)
    sa.config_set("hot_min_rows", 0)
    sa.config_set("hot_min_pct", 0)


def _case_count_sum(n, seed=11, uniform=False, shape=256):
    rng = np.random.default_rng(seed)
    x = rng.uniform(-4, 4, n) if uniform else rng.normal(0.3, 1.0, n)
    y = rng.uniform(-4, 4, n) if uniform else rng.normal(-0.5, 0.7, n)
    v = rng.normal(3, 2, n)
    v[::97] = np.nan
    x[::1013] = np.nan
    return dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4, vmax=4, bins=shape), dict(kind="scalar", data=y, vmin=-4, vmax=4, bins=shape)],
                aggs=[dict(kind="count"), dict(kind="sum", data=v), dict(kind="count", data=v)])


def test_hot_box_forced(sa, hot_pass1):
    sa.config_set("strategy", STRATEGIES["part"])
    try:
        for box in ((100, 110, 60, 50), (0, 0, 92, 92), (167, 167, 92, 92), (130, 1, 1, 200)):
            for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), box):
                sa.config_set(k, val)
            check(sa, _case_count_sum(300_000))
            assert (sa.config_get("hot_w"), sa.config_get("hot_h")) == box[2:], box
        # several chunks share one box; count(*) + sum only; sum only
        sa.config_set("part_chunk", 1 << 20)
        case = _case_count_sum(2_500_000)
        check(sa, case)
        check(sa, dict(case, aggs=case["aggs"][:2]))
        check(sa, dict(case, aggs=case["aggs"][1:2]))
        # var / std: the sum of squares (AggSumMoment, moment 2) has a plane of its own in the box; moment 3 is not the box's business
        v = case["aggs"][1]["data"]
        std_aggs = [dict(kind="count", data=v), dict(kind="sum", data=v), dict(kind="summoment", data=v, moment=2)]
        for box in ((100, 110, 60, 50), (0, 0, 70, 70)):
            for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), box):
                sa.config_set(k, val)
            check(sa, dict(case, aggs=std_aggs))
            assert sa.config_get("hot_w") == box[2]
            check(sa, dict(case, aggs=std_aggs[2:]))
            assert sa.config_get("hot_w") == box[2]
        check(sa, dict(case, aggs=std_aggs[:2] + [dict(kind="summoment", data=v, moment=3)]))
        assert sa.config_get("hot_w") == 0
        sa.config_set("hot_x0", 130); sa.config_set("hot_y0", 1); sa.config_set("hot_w", 1); sa.config_set("hot_h", 200)
        # ONE selection mask shared by every aggregator: the box stays (next to the ring-less pass 1 if that was asked for, else part_scatter_blk)
        m = case["binners"][0]["data"] > 0
        check(sa, dict(case, aggs=[dict(a, mask=m) for a in case["aggs"]]))
        assert sa.config_get("hot_w") == 1 and sa.last_kernel(0).startswith(("part_scatter_direct_hot", "part_scatter_grouped_hot", "part_scatter_phased_hot") if hot_pass1 in (3, 5, 6) else "part_scatter_hot")
        for box in ((100, 110, 60, 50), (0, 0, 92, 92)):
            for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), box):
                sa.config_set(k, val)
            check(sa, dict(case, aggs=[dict(a, mask=m) for a in case["aggs"]]))
            check(sa, dict(case, aggs=[dict(kind="count", mask=m)]))
            assert sa.config_get("hot_w") == box[2]
        # a signature the box does not serve (aggregators with different masks) runs without it
        m2 = case["binners"][1]["data"] > 0
        check(sa, dict(case, aggs=[dict(case["aggs"][0], mask=m), dict(case["aggs"][1], mask=m2)]))
        assert sa.config_get("hot_w") == 0
    finally:
        _hot_reset(sa)


def test_float32_columns_take_the_typed_kernels(sa):
    """every binner column and the value column float32 (widening on use, like BinnerScalar<float> / AggSum<float>): part_scatter_wv's
    instantiations that convert on load (round 3) or part_scatter_blk's float instantiation (no value column, the sum of squares, a
    small box), with and without a box, with a shared selection, 1-3 dims"""
    sa.config_set("strategy", STRATEGIES["part"])
    try:
        rng = np.random.default_rng(77)
        n = 1_300_001
        x, y, z = (rng.normal(0.2, 1.0, n).astype("f4") for _ in range(3))
        v = rng.normal(3, 2, n).astype("f4")
        v[::89] = np.nan
        x[::997] = np.nan
        m = v > 2.5
        b2 = [dict(kind="scalar", data=c, vmin=-4, vmax=4, bins=256) for c in (x, y)]
        aggs = [dict(kind="count"), dict(kind="sum", data=v), dict(kind="count", data=v)]
        check(sa, dict(n=n, binners=b2, aggs=aggs))
        assert sa.last_kernel(0).startswith("part_scatter") and sa.config_get("hot_w") == 0  # (below hot_min_rows: no box)
        for box in ((100, 110, 60, 50), (0, 0, 92, 92)):
            for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), box):
                sa.config_set(k, val)
            check(sa, dict(n=n, binners=b2, aggs=aggs))
            assert sa.config_get("hot_w") == box[2] and sa.last_kernel(0).startswith(("part_scatter_hot", "part_scatter_direct_hot", "part_scatter_grouped_hot", "part_scatter_phased_hot")), sa.last_kernel(0)
            check(sa, dict(n=n, binners=b2, aggs=[dict(a, mask=m) for a in aggs]))
            if box[2] * box[3] * 20 < 100_000:  # (20-byte cells with the sum of squares)
                check(sa, dict(n=n, binners=b2, aggs=[dict(kind="count", data=v), dict(kind="sum", data=v), dict(kind="summoment", data=v, moment=2)]))
                assert sa.config_get("hot_w") == box[2]
        _hot_reset(sa)
        sa.config_set("hot_min_rows", 1)
        check(sa, dict(n=n, binners=b2, aggs=aggs))          # the box from the sample (float32 sample pass)
        assert sa.config_get("hot_w") > 0 and sa.config_get("hot_fraction_ppm") > 500_000
        _hot_reset(sa)
        check(sa, dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4, vmax=4, bins=400_000)], aggs=[dict(kind="count"), dict(kind="sum", data=v)]))
        check(sa, dict(n=n, binners=[dict(kind="scalar", data=c, vmin=-4, vmax=4, bins=96) for c in (x, y, z)], aggs=[dict(kind="count", mask=m)]))
        # count(*) on float32 columns whose grid fits LDS: the float instantiation of the count kernel (plain / packed uint16 / with a selection)
        sa.config_set("strategy", 0)
        for shape, name in ((128, "count_lds_f32"), (256, "count_lds16_f32")):
            bb = [dict(kind="scalar", data=c, vmin=-4, vmax=4, bins=shape) for c in (x, y)]
            check(sa, dict(n=n, binners=bb, aggs=[dict(kind="count")]))
            assert sa.last_kernel(0) == name, sa.last_kernel(0)
            check(sa, dict(n=n, binners=bb, aggs=[dict(kind="count", mask=m)]))
        check(sa, dict(n=n, binners=[dict(kind="scalar", data=c, vmin=-4, vmax=4, bins=30) for c in (x, y, z)], aggs=[dict(kind="count")]))
        assert sa.last_kernel(0) == "count_lds_f32"
        sa.config_set("strategy", STRATEGIES["part"])
        # a float64 value column next to float32 binners is not this signature: the generic kernels
        check(sa, dict(n=n, binners=b2, aggs=[dict(kind="sum", data=v.astype("f8"))]))
    finally:
        sa.config_set("strategy", 0)
        _hot_reset(sa)


@pytest.mark.parametrize("dt,tag", [("i4", "i32"), ("i8", "i64")])
def test_integer_columns_take_the_count_kernel(sa, dt, tag):
    """count(*) binned by integer columns whose grid fits LDS: the integer instantiations of the count kernel
    (BinnerScalar<int32_t / int64_t>: the element is converted to double before the subtraction, src/binners.cpp:16-35)"""
    rng = np.random.default_rng(78)
    n = 1_100_003
    info = np.iinfo(dt)
    x = np.clip(rng.normal(0, 900, n), -5000, 5000).astype(dt)
    y = np.clip(rng.normal(100, 700, n), -5000, 5000).astype(dt)
    x[:4] = [info.min, info.max, -4000, 4000]  # far outside, and the limits themselves
    m = rng.random(n) < 0.6
    for shape, name in ((128, f"count_lds_{tag}"), (256, f"count_lds16_{tag}")):
        bb = [dict(kind="scalar", data=c, vmin=-4000, vmax=4000, bins=shape) for c in (x, y)]
        check(sa, dict(n=n, binners=bb, aggs=[dict(kind="count")]))
        assert sa.last_kernel(0) == name, sa.last_kernel(0)
        check(sa, dict(n=n, binners=bb, aggs=[dict(kind="count", mask=m)]))
        assert sa.last_kernel(0) == name
    check(sa, dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4000.5, vmax=3999.5, bins=8000)], aggs=[dict(kind="count")]))
    assert sa.last_kernel(0) == f"count_lds_{tag}"


def test_hot_box_from_sample(sa, hot_pass1):
    sa.config_set("strategy", STRATEGIES["part"])
    try:
        sa.config_set("hot_min_rows", 1)
        sa.config_set("hot_min_pct", 35)
        check(sa, _case_count_sum(3_000_000))
        assert sa.config_get("hot_w") > 0 and sa.config_get("hot_fraction_ppm") > 600_000  # sigma 1.0 x 0.7: most rows in 92x92 cells
        x0, y0, w, h = (sa.config_get(k) for k in ("hot_x0", "hot_y0", "hot_w", "hot_h"))
        assert x0 < 2 + 256 * (0.3 + 4) / 8 < x0 + w and y0 < 2 + 256 * (-0.5 + 4) / 8 < y0 + h  # the box sits on the mode
        # uniform data: no rectangle of 8.4 k cells holds 35 % of the rows -> the plain pass
        check(sa, _case_count_sum(3_000_000, uniform=True))
        assert sa.config_get("hot_w") == 0 and 0 < sa.config_get("hot_fraction_ppm") < 350_000
        # device-resident columns (the bench's route)
        # (run_superagg uploads every aggregator's input separately: two copies of v = two value columns = not the
        #  box's signature; the result must still be right)
        check(sa, _case_count_sum(3_000_000, seed=5), to_device=cases.torch_device_array)
        case = _case_count_sum(3_000_000, seed=5)
        check(sa, dict(case, aggs=case["aggs"][:2]), to_device=cases.torch_device_array)
        assert sa.config_get("hot_w") > 0
    finally:
        _hot_reset(sa)
        sa.config_set("hot_min_rows", 0)
        sa.config_set("hot_min_pct", 0)


def test_hot_box_skewed_cold_rows(sa, hot_pass1):
    # every row outside the (forced) box and in ONE cell: a tile brings 4096 records into one bucket (more than a
    # 1024-record queue block: the exact-reservation path), and with 1 Mi-row chunks the sub-queue overflows
    # (device-atomic slow path).  Then the same with the rows inside the box (nothing is emitted at all).
    sa.config_set("strategy", STRATEGIES["part"])
    try:
        for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), (100, 100, 60, 60)):
            sa.config_set(k, val)
        n = 3_000_000
        v = np.arange(n, dtype="f8") % 7
        v[::5] = np.nan
        for xv, yv in ((3.5, -3.25), (0.5, 0.25)):
            x = np.full(n, xv); y = np.full(n, yv)
            case = dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=y, vmin=-4, vmax=4, bins=256)],
                        aggs=[dict(kind="count"), dict(kind="sum", data=v), dict(kind="count", data=v)])
            for chunk in (0, 1 << 20):
                sa.config_set("part_chunk", chunk)
                got = check(sa, case)
                assert got[0].max() == n and sa.config_get("hot_w") == 60
    finally:
        _hot_reset(sa)


def test_part_blk_signatures(sa):
    # the second-generation pass 1 (part_scatter_blk) over its signature space: 1..3 dims, with / without a value
    # column, with / without a shared mask, 16..64 slabs, ragged row counts; and the same cases with blk=0
    c = cases.gaussian_columns(700_001, seed=21)
    n = 700_001
    m = c["v"] > 2.5
    vn = c["v"].copy(); vn[::53] = np.nan

    def bins(keys, shape):
        return [dict(kind="scalar", data=c[k], vmin=-4, vmax=4, bins=shape) for k in keys]

    todo = [
        dict(n=n, binners=bins("x", 500_000), aggs=[dict(kind="count")]),                                            # 1-D, 500k cells: 16 slabs
        dict(n=n, binners=bins("x", 500_000), aggs=[dict(kind="sum", data=vn, mask=m), dict(kind="count", mask=m)]),  # 1-D, value + mask
        dict(n=n, binners=bins("xy", 700), aggs=[dict(kind="count"), dict(kind="max", data=vn), dict(kind="summoment", data=vn, moment=2)]),  # 2-D 703^2, min/max + moment on the records
        dict(n=n, binners=bins("xyz", 96), aggs=[dict(kind="count", mask=m)]),                                        # 3-D 99^3 with a selection
        dict(n=n, binners=bins("xyz", 96), aggs=[dict(kind="sum", data=vn)]),                                         # 3-D with a value column
    ]
    sa.config_set("strategy", STRATEGIES["part"])
    for wv, blk in ((1, 1), (2, 1), (0, 2), (0, 0)):  # part_scatter_wv (sized / 64-record queue blocks) / part_scatter_blk / part_scatter_f64
        sa.config_set("wv", 1 if wv else 0)
        sa.config_set("wv_block", 64 if wv == 2 else 0)
        sa.config_set("blk", blk)
        try:
            used = []
            for case in todo:
                check(sa, case)
                assert sa.last_kernel(0).startswith("part_scatter")
                used.append(sa.last_kernel(0))
            if wv:  # (more than 64 slabs — the 703^2 case with three aggregators — stay on the older kernels)
                assert sum("part_scatter_wv" in k for k in used) >= 2, used
        finally:
            sa.config_set("blk", 1)
            sa.config_set("wv", WV_DEFAULT)
            sa.config_set("wv_block", 0)


def test_hot_box_many_rows_per_cell(sa, hot_pass1):
    # 4e7 rows into two neighbouring cells of the (forced) box: ~78 k rows per workgroup and cell (uint32 box counters)
    sa.config_set("strategy", STRATEGIES["part"])
    try:
        for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), (100, 100, 60, 60)):
            sa.config_set(k, val)
        n = 40_000_000
        i = np.arange(n)
        width = 8.0 / 256
        x = np.where(i % 2 == 0, 0.5 * width, 1.5 * width)  # sub-indices 130 and 131: box cells 30 and 31 of row 30
        y = np.full(n, 0.5 * width)
        v = (i % 11).astype("f8")
        v[::1000] = np.nan
        case = dict(n=n, binners=[dict(kind="scalar", data=x, vmin=-4, vmax=4, bins=256), dict(kind="scalar", data=y, vmin=-4, vmax=4, bins=256)],
                    aggs=[dict(kind="count"), dict(kind="sum", data=v), dict(kind="count", data=v)])
        got = check(sa, case)
        assert got[0].max() == n // 2 and sa.config_get("hot_w") == 60
    finally:
        _hot_reset(sa)


def test_hot_box_shared_aggregators_many_slots(sa, hot_pass1):
    # three slots (threads) feed the same aggregators chunk by chunk, every call with the (forced) box:
    # slot-private accumulators and boxes, atomic merges into the shared grids
    sa.config_set("strategy", STRATEGIES["part"])
    try:
        for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), (90, 95, 70, 66)):
            sa.config_set(k, val)
        case = _case_count_sum(2_200_000, seed=3)
        check(sa, case, chunk=300_000, nthreads=3)
        check(sa, case, chunk=1 << 20, nthreads=2, to_device=cases.torch_device_array)
    finally:
        _hot_reset(sa)
