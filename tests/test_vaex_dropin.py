"""Drop-in check with the REAL vaex Python package (only where /root/reference is mounted: this container).
vaex_amd.install() swaps `vaex.superagg`; an unmodified df.count / df.mean / df.groupby then builds OUR binners,
Grid and aggregators through vaex's own decode path (vaex/cpu.py:44-65, :630-667, vaex/agg.py:278-321 — incl.
the exact-size memory check) and reaches Grid.bin.  Without a GPU that call must fail loudly (no CPU fallback);
with one (-m gpu, if vaex is present) the results must equal the CPU reference."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVERLAY = os.path.join(ROOT, "oracle", "_ref", "overlay")
FAKE = os.path.join(ROOT, "oracle", "fake")

SCRIPT = r'''
import sys, numpy as np
sys.path[:0] = [%(overlay)r, %(fake)r, %(root)r]
import vaex, vaex_amd
cpu = vaex.superagg
hip = vaex_amd.install()
assert vaex.superagg is hip and sys.modules["vaex.superagg"] is hip and hip is not cpu
rng = np.random.default_rng(1)
n = 20000
df = vaex.from_arrays(x=rng.normal(0, 1, n), y=rng.normal(0, 1, n), v=rng.normal(3, 2, n), k=rng.integers(0, 9, n))
calls = {
  "count": lambda d: d.count(binby=["x", "y"], limits=[[-4, 4], [-4, 4]], shape=32),
  "mean": lambda d: d.mean("v", binby=["x"], limits=[-4, 4], shape=16, selection="v > 3"),
  "std": lambda d: d.std("v", binby=["x", "y"], limits=[[-4, 4], [-4, 4]], shape=8),
  "groupby": lambda d: d.groupby("k", agg={"s": vaex.agg.sum("v"), "c": vaex.agg.count()}).sort("k")["s"].to_numpy(),
}
if hip.device_count() == 0:
    for name, fn in calls.items():
        try:
            fn(df)
        except RuntimeError as e:
            assert "no HIP device" in str(e), (name, e)
            print("ok-loud-failure", name)
        else:
            raise SystemExit("computed without a GPU: " + name)
else:
    got = {name: np.asarray(fn(df)) for name, fn in calls.items()}
    vaex.superagg = cpu; sys.modules["vaex.superagg"] = cpu
    df2 = vaex.from_arrays(**{c: df[c].to_numpy() for c in ("x", "y", "v", "k")})
    for name, fn in calls.items():
        want = np.asarray(fn(df2))
        if want.dtype.kind in "iu":
            assert np.array_equal(got[name], want), name
        else:
            assert np.allclose(got[name], want, rtol=1e-10, atol=1e-12, equal_nan=True), name
        print("ok-parity", name)
'''


@pytest.mark.skipif(not os.path.isdir(OVERLAY), reason="real vaex overlay not built (needs /root/reference)")
def test_unmodified_vaex_drives_the_hip_classes():
    env = dict(os.environ, VAEX_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-c", SCRIPT % dict(overlay=OVERLAY, fake=FAKE, root=ROOT)], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok-") == 4, out.stdout
