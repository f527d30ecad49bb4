import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import vaex_amd.superagg as sa
from oracle import oracle
import importlib.util
spec = importlib.util.spec_from_file_location("t", "tests/test_gpu_first.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
ref = oracle.ref_module("superagg")
chunks = [(0, 15_000), (15_000, 15_001), (15_001, 40000)]
for seed in range(40):
    rng = np.random.default_rng(seed); n = 40000
    x, y = rng.normal(0, 1.2, n), rng.normal(0, 1.2, n)
    x[rng.random(n) < 0.01] = np.nan
    value, order = t._column(rng, "float64", n), t._column(rng, "float64", n)
    keep = rng.random(n) < 0.8
    for use_keep in (None, keep):
        wv, wm, wa = t._run(ref, x, y, value, order, use_keep, "float64", "float64", False, chunks)
        gv, gm, ga = t._run(sa, x, y, value, order, use_keep, "float64", "float64", False, chunks)
        bad = np.nonzero((gm != wm) | ((gv != wv) & ~wm))
        if len(bad[0]):
            print("seed", seed, "keep" if use_keep is not None else "nokeep", "bad cells", list(zip(*bad)))
            for c in list(zip(*bad))[:3]:
                w = np.nonzero(value == wv[c])[0]; g = np.nonzero(value == gv[c])[0]
                print("   want row", w, "order", order[w], "x,y", x[w], y[w], "keep", keep[w], "| got row", g, "order", order[g], "x,y", x[g], y[g], "keep", keep[g])
print("done")
