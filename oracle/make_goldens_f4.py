#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/vaex_api_f4.npz: the SURVEY §8 f.4 calls (first / last, nunique, groupby on
several keys, groupby with nunique, value_counts) made through the REAL vaex Python API (imported from /root/reference through
the overlay of oracle/build_ref.sh, on the reference's own compiled C++) on small seeded inputs.  Inputs AND outputs are
stored, so the GPU box replays the same calls through vaex_amd.binned.Frame (tests/test_golden_api_f4.py).

Run here (CPU container):   python oracle/make_goldens_f4.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OVERLAY = os.path.join(HERE, "_ref", "overlay")
if not os.path.isdir(OVERLAY):
    raise SystemExit("run oracle/build_ref.sh first (needs /root/reference)")
sys.path[:0] = [OVERLAY, os.path.join(HERE, "fake")]
os.environ["VAEX_NUM_THREADS"] = "1"  # (first / last / nunique have no merge in the reference — src/agg_first.cpp:42, agg_nunique.cpp:46 —: one task part)

import numpy as np  # noqa: E402
import vaex  # noqa: E402


def main():
    rng = np.random.default_rng(20260922)
    N = 6000
    x = rng.normal(0, 1, N)
    y = rng.normal(0, 1, N)
    v = rng.normal(3, 2, N)
    t = rng.permutation(N).astype("f8")            # a strict order: first / last have one answer
    q = np.round(rng.normal(0, 3, N))               # few distinct values
    qn = q.copy(); qn[rng.integers(0, N, 40)] = np.nan
    k = rng.integers(0, 40, N).astype("int64")
    k2 = (k % 5).astype("int32")
    k3 = rng.integers(-3, 4, N).astype("int64")
    i32 = rng.integers(-20, 20, N).astype("int32")
    sel = v > 3
    df = vaex.from_arrays(x=x, y=y, v=v, t=t, q=q, qn=qn, k=k, k2=k2, k3=k3, i32=i32, sel=sel)
    out = {}
    out["first_v_by_t_1d"] = df.first("v", "t", binby="x", limits=[-3, 3], shape=8)
    out["last_v_by_t_1d"] = df.last("v", "t", binby="x", limits=[-3, 3], shape=8)
    out["first_i32_by_t_2d"] = df.first("i32", "t", binby=["x", "y"], limits=[[-3, 3], [-3, 3]], shape=4)
    out["first_v_by_t_1d_sel"] = df.first("v", "t", binby="x", limits=[-3, 3], shape=8, selection="sel")
    out["nunique_q_1d"] = df._compute_agg("nunique", "q", binby="y", limits=[-3, 3], shape=8)
    out["nunique_qn_1d"] = df._compute_agg("nunique", "qn", binby="y", limits=[-3, 3], shape=8)      # the NaN counts as one value
    out["nunique_i32_2d"] = df._compute_agg("nunique", "i32", binby=["x", "y"], limits=[[-3, 3], [-3, 3]], shape=4)
    out["nunique_q_1d_sel"] = df._compute_agg("nunique", "q", binby="y", limits=[-3, 3], shape=8, selection="sel")
    out["nunique_q_scalar"] = np.array(df._compute_agg("nunique", "q"))
    g = df.groupby(["k2", "k3"], agg={"c": vaex.agg.count(), "s": vaex.agg.sum("v"), "m": vaex.agg.mean("v")}).sort(["k2", "k3"])
    for col in ("k2", "k3", "c", "s", "m"):
        out[f"groupby2_{col}"] = g[col].to_numpy()
    g = df.groupby("k", agg={"u": vaex.agg.nunique("i32"), "uq": vaex.agg.nunique("qn")}).sort("k")
    for col in ("k", "u", "uq"):
        out[f"groupby_nunique_{col}"] = g[col].to_numpy()
    vc = df.k.value_counts()
    out["value_counts_k_values"] = np.asarray(vc.index)
    out["value_counts_k_counts"] = np.asarray(vc.values)
    vc = df.qn.value_counts(dropna=False)
    out["value_counts_qn_values"] = np.asarray(vc.index, dtype="f8")
    out["value_counts_qn_counts"] = np.asarray(vc.values)
    inputs = dict(x=x, y=y, v=v, t=t, q=q, qn=qn, k=k, k2=k2, k3=k3, i32=i32, sel=sel)
    path = os.path.join(ROOT, "tests", "golden", "vaex_api_f4.npz")
    np.savez_compressed(path, **{"in_" + k_: a for k_, a in inputs.items()},
                        **{"out_" + k_: (np.ma.getdata(a) if np.ma.isMaskedArray(a) else np.asarray(a)) for k_, a in out.items()},
                        **{"mask_" + k_: np.ma.getmaskarray(a) for k_, a in out.items() if np.ma.isMaskedArray(a)})
    print("wrote", path, "with", len(out), "results")
    for k_, a in out.items():
        a = np.asarray(a)
        print(f"  {k_:<28} {str(a.dtype):<8} {a.shape}")


if __name__ == "__main__":
    main()
