"""Replays the SURVEY §8 f.4 calls of oracle/make_goldens_f4.py (made there through the REAL vaex API: df.first / df.last,
nunique, groupby on two keys, groupby with nunique, value_counts) through vaex_amd.binned.Frame and compares with the committed
fixture tests/golden/vaex_api_f4.npz:
  * CPU: Frame driving the reference's compiled AggFirst / AggNUnique classes (oracle/_ref) — Frame's host logic for them;
  * GPU (-m gpu): Frame driving the HIP path (vxh_first_*, vxh_collect_*, vxh_pack_keys + groupby) through the C-ABI."""
import os

import numpy as np
import pytest

from tests.conftest import knob
from tests.test_golden_api import RefAdapter

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "vaex_api_f4.npz")


def load():
    z = np.load(GOLDEN)
    cols = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    masks = {k[5:]: z[k] for k in z.files if k.startswith("mask_")}
    return cols, out, masks


def _first_with_selection(df, out, masks, cols, like_the_reference):
    """df.first(..., selection=): vaex hands the selection to AggFirst as its keep-mask in ONE call of all 6000 rows, and the
    reference reads that mask at the row's position inside the current 1024-row block (src/agg_first.cpp:131) — the fixture
    holds that answer.  `like_the_reference`: the frame reproduces it (one call + the reference's indexing); else it must
    give what the call means: per cell the value of the selected row with the smallest t."""
    name = "first_v_by_t_1d_sel"
    got = df.first("v", "t", binby="x", limits=[-3, 3], shape=8, selection="sel")
    if like_the_reference:
        m = masks.get(name, np.zeros(out[name].shape, bool))
        assert np.array_equal(np.ma.getmaskarray(got), m) and np.array_equal(np.ma.getdata(got)[~m], out[name][~m])
        return
    x, v, t, sel = (np.asarray(cols[k].cpu() if hasattr(cols[k], "cpu") else cols[k]) for k in ("x", "v", "t", "sel"))
    cell = np.floor((x + 3) / 6 * 8)
    for c in range(8):
        rows = np.nonzero((cell == c) & sel.astype(bool))[0]
        assert np.ma.getdata(got)[c] == v[rows[np.argmin(t[rows])]]


def _first_last_and_nunique(df, out, masks):
    lim1, lim2 = [-3, 3], [[-3, 3], [-3, 3]]
    for name, got in (("first_v_by_t_1d", df.first("v", "t", binby="x", limits=lim1, shape=8)),
                      ("last_v_by_t_1d", df.last("v", "t", binby="x", limits=lim1, shape=8)),
                      ("first_i32_by_t_2d", df.first("i32", "t", binby=["x", "y"], limits=lim2, shape=4))):
        m = masks.get(name, np.zeros(out[name].shape, bool))
        assert np.array_equal(np.ma.getmaskarray(got), m), name
        assert np.array_equal(np.ma.getdata(got)[~m], out[name][~m]), name
        assert np.ma.getdata(got).dtype == out[name].dtype, name
    for name, got in (("nunique_q_1d", df.nunique("q", binby="y", limits=lim1, shape=8)),
                      ("nunique_qn_1d", df.nunique("qn", binby="y", limits=lim1, shape=8)),
                      ("nunique_i32_2d", df.nunique("i32", binby=["x", "y"], limits=lim2, shape=4)),
                      ("nunique_q_1d_sel", df.nunique("q", binby="y", limits=lim1, shape=8, selection="sel")),
                      ("nunique_q_scalar", np.array(df.nunique("q")))):
        assert np.array_equal(np.asarray(got), out[name]), (name, got, out[name])
    # df.groupby("k", agg={"u": nunique(i32), "uq": nunique(qn)}): the 40 dense keys bin themselves
    assert np.array_equal(out["groupby_nunique_k"], np.arange(40))
    assert np.array_equal(df.nunique("i32", binby=[dict(column="k", count=40)]), out["groupby_nunique_u"])
    assert np.array_equal(df.nunique("qn", binby=[dict(column="k", count=40)]), out["groupby_nunique_uq"])


def test_f4_goldens_frame_on_reference_cpp(ref):
    from vaex_amd.binned import Frame
    cols, out, masks = load()
    _first_last_and_nunique(Frame(cols, chunk_size=1000, nthreads=1, superagg=RefAdapter(ref)), out, masks)
    # the selection case: one call of all rows reproduces the fixture (the reference's block-local mask index); calls of
    # <= 1024 rows, where the reference reads its keep-mask at the right rows, give what the call means
    _first_with_selection(Frame(cols, chunk_size=len(cols["x"]), nthreads=1, superagg=RefAdapter(ref)), out, masks, cols, True)
    _first_with_selection(Frame(cols, chunk_size=1000, nthreads=1, superagg=RefAdapter(ref)), out, masks, cols, False)


@pytest.mark.gpu
@pytest.mark.parametrize("device", [False, True])
def test_f4_goldens_frame_on_hip(sa, gpu_ready, device):
    from vaex_amd.binned import Frame, agg
    cols, out, masks = load()
    if device:
        import torch
        cols = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cols.items()}
    df = Frame(cols, chunk_size=1000, nthreads=3)
    _first_last_and_nunique(df, out, masks)
    if not device:   # host columns go in 1000-row calls, where the reference's block-local mask index IS the row's (device columns are binned in place, whole)
        _first_with_selection(df, out, masks, cols, False)
    # one call of all rows with the library's default (the reference's block-local indexing): the fixture
    assert sa.config_get("first_mask_block") == 1024
    _first_with_selection(Frame(cols, chunk_size=len(cols["x"]), nthreads=1), out, masks, cols, True)
    with knob(sa, "first_mask_block", 0):               # mask[row]: what the call means, whatever the chunking
        _first_with_selection(Frame(cols, chunk_size=len(cols["x"]), nthreads=1), out, masks, cols, False)
    # groupby on two keys (GrouperCombined, vaex/groupby.py:526-584), sorted by (k2, k3) like the fixture
    g = df.groupby(["k2", "k3"], {"c": agg.count(), "s": agg.sum("v"), "m": agg.mean("v")})
    assert np.array_equal(g["k2"], out["groupby2_k2"]) and np.array_equal(g["k3"], out["groupby2_k3"])
    assert np.array_equal(g["c"], out["groupby2_c"])
    assert np.allclose(g["s"], out["groupby2_s"], rtol=1e-12, atol=1e-12 * 20 * out["groupby2_c"].max())
    assert np.allclose(g["m"], out["groupby2_m"], rtol=1e-12, atol=1e-12)
    # value_counts: the same (value, count) pairs — values told apart by their bits (-0.0 and +0.0 are two entries in the
    # reference's counter as well) —, counts descending (ties: vaex leaves their order to pandas)
    def pairs(values, counts):
        v = np.asarray(values, dtype="f8")
        return sorted((2**63 if np.isnan(a) else int(np.float64(a).view("i8")), int(c)) for a, c in zip(v.tolist(), np.asarray(counts).tolist()))
    for col in ("k", "qn"):
        vals, counts = df.value_counts(col)
        assert np.array_equal(np.sort(counts)[::-1], counts)
        assert pairs(vals, counts) == pairs(out[f"value_counts_{col}_values"], out[f"value_counts_{col}_counts"])
