"""SURVEY.md §8 row a12 — the legacy vaexfast.statisticNd_f8 pass.
CPU: the C restatement (oracle.statistic_nd) is pinned against the reference's own compiled vaexfast
(oracle/_ref) for every op / dimensionality / edges mode.  GPU: vaex_amd.vaexfast.statisticNd_f8 (HIP
kernels through the C-ABI) against the restatement: counts and min/max exact, fp64 sums within 1e-12."""
import numpy as np
import pytest

from oracle import oracle

OPS = {0: "add1", 1: "count", 2: "min_max", 3: "moments_01", 4: "moments_012"}


def make_case(seed, nd, n, op, use_edges, sizes=None):
    rng = np.random.default_rng(seed)
    sizes = sizes or [7, 5, 4, 6][:nd]
    blocks = []
    for d in range(nd):
        b = rng.normal(0.0, 1.5, n)
        b[rng.random(n) < 0.02] = np.nan
        blocks.append(b)
    w = rng.normal(1.0, 3.0, n)
    w[rng.random(n) < 0.05] = np.nan
    minima = [-2.0 - 0.25 * d for d in range(nd)]
    maxima = [2.5 + 0.5 * d for d in range(nd)]
    fields = oracle.STAT_FIELDS[op]
    grid = np.zeros(tuple(sizes) + (fields,), dtype=np.float64)
    if op == 2:
        grid[..., 0] = np.inf
        grid[..., 1] = -np.inf
    weights = None if op == 0 else [w]
    return blocks, weights, grid, minima, maxima


CASES = [(nd, op, e) for nd in (0, 1, 2, 3, 4) for op in OPS for e in (0, 1)]


@pytest.mark.parametrize("nd,op,use_edges", CASES)
def test_restatement_pinned_to_reference_vaexfast(nd, op, use_edges):
    vf = oracle.ref_module("vaexfast")
    if vf is None:
        pytest.skip("oracle/_ref/vaexfast not built (reference sources absent)")
    blocks, weights, grid, minima, maxima = make_case(100 + nd * 10 + op, nd, 5000, op, use_edges)
    want = grid.copy()
    got = grid.copy()
    # two chunks: the grid is accumulated into, not overwritten
    for i1, i2 in ((0, 2000), (2000, 5000)):
        bs = [b[i1:i2] for b in blocks]
        ws = None if weights is None else [w[i1:i2] for w in weights]
        vf.statisticNd_f8(bs, ws, want, [float(m) for m in minima], [float(m) for m in maxima], op, use_edges)
        oracle.statistic_nd(bs, ws, got, minima, maxima, op, use_edges)
    np.testing.assert_array_equal(got, want)  # same scalar loop, same order: bit-exact
    if op == 0 and not use_edges and nd:
        assert 0 < want.sum() < 5000


def compare(got, want, op):
    if op in (0, 1, 2):
        np.testing.assert_array_equal(got, want)
    else:
        np.testing.assert_array_equal(got[..., 0], want[..., 0])
        # fp64 sums: a different accumulation order on the GPU; 1e-12 relative to sum|w| (|w| < 20, <= 2e5 rows)
        for f in range(1, got.shape[-1]):
            scale = max(1.0, np.abs(want[..., f]).max())
            np.testing.assert_allclose(got[..., f], want[..., f], rtol=1e-12, atol=1e-12 * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("nd,op,use_edges", CASES)
def test_statistic_nd_hip_vs_oracle(sa, gpu_ready, nd, op, use_edges):
    from vaex_amd import vaexfast
    blocks, weights, grid, minima, maxima = make_case(300 + nd * 10 + op, nd, 20000, op, use_edges)
    want, got = grid.copy(), grid.copy()
    for i1, i2 in ((0, 7000), (7000, 20000)):
        bs = [b[i1:i2] for b in blocks]
        ws = None if weights is None else [w[i1:i2] for w in weights]
        oracle.statistic_nd(bs, ws, want, minima, maxima, op, use_edges)
        assert vaexfast.statisticNd_f8(bs, ws, got, minima, maxima, op, use_edges) is None
    compare(got, want, op)


@pytest.mark.gpu
def test_statistic_nd_device_blocks_and_big_endian(sa, gpu_ready):
    import torch
    from vaex_amd import vaexfast
    blocks, weights, grid, minima, maxima = make_case(77, 2, 200000, 4, 0, sizes=[64, 32])
    want = grid.copy()
    oracle.statistic_nd(blocks, weights, want, minima, maxima, 4, 0)
    got = grid.copy()
    vaexfast.statisticNd_f8([torch.from_numpy(b).cuda() for b in blocks], [torch.from_numpy(weights[0]).cuda()], got, minima, maxima, 4)
    compare(got, want, 4)
    got = grid.copy()
    vaexfast.statisticNd_f8([b.astype(">f8") for b in blocks], [weights[0].astype(">f8")], got, minima, maxima, 4)
    compare(got, want, 4)


@pytest.mark.gpu
def test_statistic_nd_rejects(sa, gpu_ready):
    from vaex_amd import vaexfast
    g = np.zeros((4, 1))
    x = np.zeros(8)
    with pytest.raises(NotImplementedError):
        vaexfast.statisticNd_f8([x], [x, x], np.zeros((4, 8)), [0.0], [1.0], vaexfast.OP_COV)
    with pytest.raises(TypeError):
        vaexfast.statisticNd_f8([x.astype("f4")], None, g, [0.0], [1.0], 0)
    with pytest.raises(ValueError):
        vaexfast.statisticNd_f8([x], None, np.zeros((4, 4, 1)), [0.0], [1.0], 0)
    with pytest.raises(RuntimeError):
        vaexfast.statisticNd_f8([x], None, np.zeros((4, 2))[:, ::2], [0.0], [1.0], 0)
    with pytest.raises(ValueError):
        vaexfast.statisticNd_f8([x], None, g, [0.0], [1.0], 1)
