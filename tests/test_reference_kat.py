"""Known-answer vectors of the reference's OWN tests for this path, replayed through vaex_amd.binned.Frame:
on the reference's compiled C++ (CPU: Frame's host logic) and on the HIP path (-m gpu).  Each case cites the reference test
it restates; the expected values are the literals of those tests (or numpy / pandas exactly as those tests compute them)."""
import numpy as np
import pytest

from tests.test_golden_api import RefAdapter


def _frames(request_gpu, ref=None, sa=None):
    from vaex_amd.binned import Frame
    if request_gpu:
        return lambda **cols: Frame(cols, chunk_size=4, nthreads=2)
    return lambda **cols: Frame(cols, chunk_size=4, nthreads=1, superagg=RefAdapter(ref))


def _cases(make, gpu):
    from vaex_amd.binned import agg
    # tests/groupby_test.py:116-122 test_groupby_1d
    g = np.array([0, 0, 0, 0, 1, 1, 1, 1, 2, 2])
    r = make(g=g, x=np.arange(10.0)).groupby("g", {"count": agg.count()})
    assert r["g"].tolist() == [0, 1, 2] and r["count"].tolist() == [4, 4, 2]
    # tests/groupby_test.py:426-441 test_groupby_std (the string labels '0','1','2' as integers): pandas std(ddof=0)
    gv = np.array([9, 2, 3, 4, 0, 1, 2, 3, 2, 5], dtype="int32")
    s = np.array([0, 0, 0, 0, 1, 1, 1, 1, 2, 2])
    r = make(g=gv, s=s).groupby("s", {"g": agg.std("g")})
    want = [np.std(gv[s == k].astype("f8")) for k in range(3)]
    assert r["s"].tolist() == [0, 1, 2]
    np.testing.assert_array_almost_equal(r["g"], want)
    # tests/groupby_test.py:619-636 test_binner_2d: Binner(x, 0, 3, bins=3) x Grouper(g) -> xarray data [[1, 0], [1, 1], [1, 2]]
    x = np.array([0.1, 1.1, 1.2, 2.2, 2.5, 2.7, 100])
    gg = np.array([0, 0, 1, 0, 1, 1, 1])
    c = make(x=x, g=gg).count(binby=["x", dict(column="g", count=2)], limits=[[0, 3], None], shape=[3, 2])
    assert np.asarray(c).tolist() == [[1, 0], [1, 1], [1, 2]]
    # tests/count_test.py:43-50 (test_count_edges): 1 missing, 2 to the left, 1 in the range, 2 to the right
    xe = np.array([-2, -1, 0, 1, 2, 3, np.nan])
    f = make(x=xe)
    assert f.count(binby="x", limits=[0.5, 1.5], shape=1, edges=True).tolist() == [1, 3, 1, 2]
    assert f.count("x", binby="x", limits=[0.5, 1.5], shape=1, edges=True).tolist() == [0, 3, 1, 2]  # the NaN value itself is not counted
    assert f.count("x", binby="x", limits=[0.5, 1.5], shape=1, edges=False).tolist() == [1]
    xm = np.ma.array(np.array([-2, -1, 0, 1, 2, 3, 4]), mask=np.arange(7) == 6)
    assert make(x=xm).count(binby="x", limits=[0.5, 1.5], shape=1, edges=True).tolist() == [1, 3, 1, 2]
    # tests/count_test.py:26-41 test_count_1d_verify_against_numpy, limits='minmax' (the ds_local fixture's x = arange(10), y = x**2)
    x10 = np.arange(10.0)
    y10 = x10 ** 2
    sel = y10 > 10
    f = make(x=x10, y=y10)
    lo, hi = f.minmax("x", selection=sel)
    counts = f.count(binby=["x"], selection=sel, shape=4, limits=[[lo, hi]])
    np_counts, _ = np.histogram(x10[sel], bins=4, range=(lo, hi))
    assert counts[:-1].tolist() == np_counts[:-1].tolist()
    # tests/agg_test.py:150-158 (1-d count golden: also pinned at the class level in tests/test_gpu_parity.py)
    xa = np.array([-1, -2, 0.5, 1.5, 4.5, 5], dtype="f8")
    assert make(x=xa).count(binby="x", limits=[0, 5], shape=5, edges=True).tolist() == [0, 2, 1, 1, 0, 0, 1, 1]
    # tests/agg_test.py:108-132 test_count_basics (fixture: x = arange(10), y = x**2)
    x = np.arange(10.0)
    y = x ** 2
    f = make(x=x, y=y)
    counts = f.count(binby="x", limits=[0, 10], shape=10)
    assert len(counts) == 10 and all(counts == 1)
    assert all(f.sum("y", binby="x", limits=[0, 10], shape=10) == y)
    mask = x < 5
    counts = f.count("x", binby="x", limits=[0, 10], shape=10, selection=mask)
    assert all(counts == mask * 1)
    assert all(f.sum("y", binby="x", limits=[0, 10], shape=10, selection=mask) == np.where(mask, y, 0))
    # tests/agg_test.py:8-48 test_sum, the binned half: x with a NaN in row 0, selection x < 5
    xn = x.copy()
    xn[0] = np.nan
    f = make(x=xn, y=y)
    sel5 = np.arange(10) < 5
    A = np.testing.assert_array_almost_equal
    A(f.sum("x", binby=["y"], limits=[0, 9 ** 2 + 1], shape=1), [np.nansum(xn)])
    A(f.sum("x", binby=["y"], limits=[0, 9 ** 2 + 1], shape=1, selection=sel5), [np.nansum(xn[:5])])
    A(f.sum("x", binby=["y"], limits=[0, 9 ** 2 + 1], shape=2), [np.nansum(xn[:7]), np.nansum(xn[7:])])
    A(f.sum("x", binby=["y"], limits=[0, 9 ** 2 + 1], shape=2, selection=sel5), [np.nansum(xn[:5]), 0])
    A(f.sum("y", binby=["x"], limits=[0, 10], shape=2), [np.nansum(y[1:5]), np.nansum(y[5:])])   # (row 0: x is NaN — the NaN cell)
    A(f.sum("y", binby=["x"], limits=[0, 10], shape=2, selection=sel5), [np.nansum(y[1:5]), 0])
    # tests/agg_test.py:171-180 test_count_1d_ordinal: ordinal binner of 5, count(edges=True)
    xo = np.array([-1, -2, 0, 1, 4, 5], dtype="i8")
    assert make(x=xo).count(binby=[dict(column="x", count=5)], edges=True).tolist() == [1, 1, 0, 0, 1, 3, 0]
    # tests/agg_test.py:184-192 test_mean_basics (x = arange(10), y = x**2)
    f = make(x=x, y=y)
    assert float(f.mean("x")) == 4.5 and float(f.mean("y")) == 28.5
    assert float(f.mean("x", selection=x < 3)) == 1 and float(f.mean("y", selection=x < 3)) == 5 / 3
    # tests/agg_test.py:257-262 test_big_endian_binning, :265-273 non-contiguous big-endian, :276-282 test_strides
    xb = np.arange(10, dtype=">f8")
    yb = np.zeros(10, dtype=">f8")
    counts = make(x=xb, y=yb).count(binby=["x", "y"], limits=[[-0.5, 9.5], [-0.5, 0.5]], shape=[10, 1])
    assert counts.ravel().tolist() == np.ones(10).tolist()
    xs = np.arange(20, dtype=">f8")[::2]
    xs[:] = np.arange(10, dtype=">f8")
    ys = np.arange(20, dtype=">f8")[::2]
    ys[:] = np.arange(10, dtype=">f8")
    counts = make(x=xs, y=ys).count(binby=["x", "y"], limits=[[-0.5, 9.5], [-0.5, 9.5]], shape=[10, 10])
    assert np.diagonal(counts).tolist() == np.ones(10).tolist()
    ar = np.zeros((10, 2)).reshape(20)
    xst = ar[::2]
    xst[:] = np.arange(10)
    assert make(x=xst).count(binby="x", limits=[-0.5, 9.5], shape=10).tolist() == np.ones(10).tolist()
    if gpu:
        # tests/agg_test.py:294-316 test_nunique, float half (AggNUnique on the HIP path)
        mapping = {"aap": 1.2, "noot": 2.5, "mies": 3.7, "kees": 4.8, None: np.nan}
        sv = np.array([mapping[k] for k in ["aap", "aap", "noot", "mies", None, "mies", "kees", "mies", "aap"]], dtype="f8")
        xg = np.array([0, 0, 0, 0, 0, 1, 1, 1, 2])
        f = make(x=xg, s=sv)
        assert f.nunique("s", binby=[dict(column="x", count=3)]).tolist() == [4, 2, 1]
        assert f.nunique("s", binby=[dict(column="x", count=3)], dropnan=True).tolist() == [3, 2, 1]


def test_reference_known_answers_on_reference_cpp(ref):
    _cases(_frames(False, ref=ref), False)


@pytest.mark.gpu
def test_reference_known_answers_on_hip(sa, gpu_ready):
    _cases(_frames(True), True)
