#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — generates tests/golden/vaex_api.npz by running the REAL vaex Python API
(imported from /root/reference through the overlay that oracle/build_ref.sh creates, on top of the
reference's own compiled C++) on small seeded inputs.  The fixture stores inputs AND outputs, so the GPU
box (where /root/reference does not exist) can replay the same calls through vaex_amd.binned.Frame.

Run here (CPU container):   python oracle/make_goldens.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OVERLAY = os.path.join(HERE, "_ref", "overlay")
if not os.path.isdir(OVERLAY):
    raise SystemExit("run oracle/build_ref.sh first (needs /root/reference)")
sys.path[:0] = [OVERLAY, os.path.join(HERE, "fake")]
os.environ.setdefault("VAEX_NUM_THREADS", "4")

import numpy as np  # noqa: E402
import vaex  # noqa: E402


def main():
    rng = np.random.default_rng(20260921)
    N = 4000
    x = rng.normal(0, 1, N)
    y = rng.normal(0, 1, N)
    z = rng.normal(0, 1, N)
    v = rng.normal(3, 2, N)
    v[rng.integers(0, N, 25)] = np.nan
    x[:3] = [-np.inf, np.inf, 4.0]
    x[rng.integers(3, N, 10)] = np.nan
    k = rng.integers(0, 50, N).astype("int64")
    k[k == 17] = 18  # a hole in the dense range
    ks = k * 7919 + 3  # sparse keys -> hash path
    i32 = rng.integers(-1000, 1000, N).astype("int32")
    u8 = rng.integers(0, 255, N).astype("uint8")
    f32 = rng.normal(0, 1, N).astype("float32")
    xb = x.astype(">f8")
    sel = v > 3
    mvals = rng.normal(0, 1, N)
    mmask = rng.random(N) < 0.2
    m = np.ma.array(mvals, mask=mmask)

    df = vaex.from_arrays(x=x, y=y, z=z, v=v, k=k, ks=ks, i32=i32, u8=u8, f32=f32, xb=xb, sel=sel, m=m)
    lim2 = [[-4, 4], [-4, 4]]
    out = {}
    out["count_2d"] = df.count(binby=["x", "y"], limits=lim2, shape=16)
    out["count_2d_edges"] = df.count(binby=["x", "y"], limits=lim2, shape=16, edges=True)
    out["count_v_2d"] = df.count("v", binby=["x", "y"], limits=lim2, shape=16)
    out["count_2d_sel"] = df.count(binby=["x", "y"], limits=lim2, shape=16, selection="sel")
    out["sum_v_2d"] = df.sum("v", binby=["x", "y"], limits=lim2, shape=16)
    out["mean_v_2d"] = df.mean("v", binby=["x", "y"], limits=lim2, shape=16)
    out["mean_v_2d_sel"] = df.mean("v", binby=["x", "y"], limits=lim2, shape=16, selection="sel")
    out["var_v_2d"] = df.var("v", binby=["x", "y"], limits=lim2, shape=8)
    out["std_v_2d"] = df.std("v", binby=["x", "y"], limits=lim2, shape=8)
    out["min_v_1d"] = df.min("v", binby="x", limits=[-3, 3], shape=8)
    out["max_v_1d"] = df.max("v", binby="x", limits=[-3, 3], shape=8)
    out["minmax_y"] = df.minmax("y")
    out["count_1d_limits_none"] = df.count(binby="y", shape=8)
    out["sum_i32_1d"] = df.sum("i32", binby="y", limits=[-3, 3], shape=8)
    out["sum_u8_1d"] = df.sum("u8", binby="y", limits=[-3, 3], shape=8)
    out["sum_f32_1d"] = df.sum("f32", binby="y", limits=[-3, 3], shape=8)
    out["std_i32_1d"] = df.std("i32", binby="y", limits=[-3, 3], shape=8)
    out["count_bigendian_1d"] = df.count(binby="xb", limits=[-3, 3], shape=8)
    out["mean_bigendian_1d"] = df.mean("xb", binby="y", limits=[-3, 3], shape=8)
    out["mean_masked_1d"] = df.mean("m", binby="y", limits=[-3, 3], shape=8)
    out["count_masked_binby_1d"] = df.count(binby="m", limits=[-3, 3], shape=8, edges=True)
    out["count_3d"] = df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=6)
    out["count_f32_binby"] = df.count(binby="f32", limits=[-3, 3], shape=8)
    out["count_i32_binby"] = df.count(binby="i32", limits=[-1000, 1000], shape=10)
    out["sum_scalar"] = np.array(df.sum("v"))
    out["count_scalar"] = np.array(df.count())
    # limits from the data — a min/max pass + a 1-d count pass + numpy (dataframe.py:1795-1840, :1632-1760)
    out["limits_pct_y_90"] = df.limits_percentage("y", 90)
    out["limits_pct_v_default"] = df.limits_percentage("v")
    out["limits_pct_y_sel"] = df.limits_percentage("y", 95, selection="sel")
    out["percentile_y_50"] = df.percentile_approx("y", 50)
    out["percentile_y_multi"] = df.percentile_approx("y", [0, 10, 25, 50, 99, 100])
    out["percentile_v_by_y"] = df.percentile_approx("v", 50, binby=["y"], limits=[[-3, 3]], shape=6)
    out["median_y_sel"] = df.median_approx("y", selection="sel")
    out["mean_scalar"] = np.array(df.mean("v"))

    for name, key in (("dense", "k"), ("sparse", "ks")):
        g = df.groupby(key, agg={"c": vaex.agg.count(), "s": vaex.agg.sum("v"), "m": vaex.agg.mean("v"), "sd": vaex.agg.std("v"), "mn": vaex.agg.min("v"), "mx": vaex.agg.max("v")}).sort(key)
        out[f"groupby_{name}_keys"] = g[key].to_numpy()
        for col in ("c", "s", "m", "sd", "mn", "mx"):
            out[f"groupby_{name}_{col}"] = g[col].to_numpy()

    inputs = dict(x=x, y=y, z=z, v=v, k=k, ks=ks, i32=i32, u8=u8, f32=f32, sel=sel, mvals=mvals, mmask=mmask)
    path = os.path.join(ROOT, "tests", "golden", "vaex_api.npz")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, **{"in_" + k_: a for k_, a in inputs.items()}, **{"out_" + k_: np.asarray(a) for k_, a in out.items()})
    print("wrote", path, "with", len(out), "results; vaex", vaex.__version__ if hasattr(vaex, "__version__") else "")
    for k_, a in out.items():
        a = np.asarray(a)
        print(f"  {k_:<28} {str(a.dtype):<8} {a.shape}")


if __name__ == "__main__":
    main()
