"""Round 6: what a PROCESS pays on its first 1e9-row groupby / 3-D histogram (0.64-0.97 s on the driver's line, `ms_first_call_in_process`).
Fresh subprocesses: the big call first / a tiny call of the same kind first (code objects, streams, small buffers), then the big one (what is left is allocation);
the block pool's hipMalloc clock and the host clock of every library call are printed.
    python tools/r06_process_first.py [rows=1e9] [kinds=scattered,dense,c3d] [warm=0,1,2]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, time
t_start = time.perf_counter()
import numpy as np, torch
sys.path.insert(0, %(root)r)
import vaex_amd
from vaex_amd.binned import Frame, agg
sa = vaex_amd.superagg
rows, kind, warm = %(rows)d, %(kind)r, %(warm)d
t_import = time.perf_counter() - t_start
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter()
sa.device_count(); sa.synchronize()
t_libinit = time.perf_counter() - t0
warm_ms = None
if warm == 2 and hasattr(sa, "warmup"):
    t0 = time.perf_counter(); sa.warmup(); warm_ms = (time.perf_counter() - t0) * 1e3
g = torch.Generator(device="cuda").manual_seed(7)
def data(n):
    v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    if kind == "c3d":
        return dict(x=torch.randn(n, dtype=torch.float64, device="cuda", generator=g), y=torch.randn(n, dtype=torch.float64, device="cuda", generator=g),
                    z=torch.randn(n, dtype=torch.float64, device="cuda", generator=g), sel=(v > 3).to(torch.uint8))
    k = torch.randint(0, 1_000_000, (n,), dtype=torch.int64, device="cuda", generator=g)
    if kind == "scattered":
        k = (k * 2654435761) %% (1 << 40)
    return dict(k=k, v=v)
spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
def call(cols):
    names = [nm for nm in ("groupby_run", "scan_key_value", "minmax_int", "minmax", "finish") if hasattr(sa, nm)]
    saved, log = {nm: getattr(sa, nm) for nm in names}, []
    def wrap(nm, f):
        def g_(*a, **kw):
            tw = time.perf_counter()
            try:
                return f(*a, **kw)
            finally:
                log.append([nm, round((time.perf_counter() - tw) * 1e3, 2)])
        return g_
    pool0 = {k_: sa.config_get(k_) for k_ in ("pool_mallocs", "pool_malloc_bytes", "pool_malloc_us")}
    for nm in names: setattr(sa, nm, wrap(nm, saved[nm]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f = Frame(cols)
    r = f.groupby("k", spec) if kind != "c3d" else f.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128, selection="sel", edges=True)
    ms = (time.perf_counter() - t0) * 1e3
    for nm in names: setattr(sa, nm, saved[nm])
    return {"ms": round(ms, 2), "calls": log, "pool": {k_: sa.config_get(k_) - v_ for k_, v_ in pool0.items()}}
out = {"kind": kind, "warm": warm, "import_s": round(t_import, 2), "libinit_ms": round(t_libinit * 1e3, 2), "warmup_ms": warm_ms}
if warm == 1:
    out["tiny_first"] = call(data(200_000))
cols = data(rows)
torch.cuda.synchronize()
out["big_first"] = call(cols)
out["big_second_fresh_columns"] = call({k_: c.clone() for k_, c in cols.items()})
out["big_third_same_sizes"] = call({k_: c.clone() for k_, c in cols.items()})
print("RESULT " + json.dumps(out))
'''


def main():
    rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
    kinds = (sys.argv[2] if len(sys.argv) > 2 else "scattered,dense,c3d").split(",")
    warms = [int(w) for w in (sys.argv[3] if len(sys.argv) > 3 else "0,1,2").split(",")]
    for kind in kinds:
        for warm in warms:
            p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, rows=rows, kind=kind, warm=warm)], capture_output=True, text=True, timeout=600,
                               env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd="/tmp")
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            print(line[0][7:] if line else json.dumps({"kind": kind, "warm": warm, "rc": p.returncode, "stderr": p.stderr[-1500:]}), flush=True)


if __name__ == "__main__":
    main()
