import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import vaex_amd
from oracle import oracle
from tests import cases
sa = vaex_amd.superagg
sa.config_set("strategy", 4)
for n in (1, 63, 63, 1000):
    case = cases.case_2d_count_mean(n, shape=32)
    want = oracle.run_case(case)
    got = cases.run_superagg(sa, case)
    for k in range(3):
        g, w = np.asarray(got[k]), np.asarray(want[k])
        bad = np.argwhere(~np.isclose(g, w, rtol=1e-12, atol=0))
        print(f"n={n} agg{k}: sum got {g.sum()} want {w.sum()} mismatches {len(bad)}", flush=True)
        for b in bad[:6]:
            print("    cell", tuple(b), "flat", b[0] + 35 * b[1], "got", g[tuple(b)], "want", w[tuple(b)])
    print(sa.last_kernel(0))
