"""Round 6 diagnosis: the driver's box showed 463.8 ms for the FIRST call of the scattered 1e6-key groupby over fresh columns
(bench.py configs[3]': prime on clones, torch.cuda.empty_cache(), first call) where the builder's boxes show 14 ms.  The same sequence, many
times, with the host-side phases of every call on the clock (key / NaN scan, heavy-key sample, the fused pass, result columns) next to the pass's own
info (retries, buckets, kernel times).
    python tools/r06_first_call.py [rows=1e9] [rounds=12]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import vaex_amd
from vaex_amd.binned import Frame, agg

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12


class Timed:
    """vaex_amd.superagg with a host clock around the calls a groupby makes"""

    def __init__(self):
        self.log = []

    def __getattr__(self, name):
        f = getattr(sa, name)
        if name in ("groupby_run", "scan_key_value", "minmax_int", "finish"):
            def wrapped(*a, **kw):
                t0 = time.perf_counter()
                r = f(*a, **kw)
                self.log.append((name, round((time.perf_counter() - t0) * 1e3, 3)))
                return r
            return wrapped
        return f


def call(frame, spec, proxy):
    proxy.log.clear()
    heavy = Frame._heavy_keys

    def timed_heavy(self, *a, **kw):
        t0 = time.perf_counter()
        r = heavy(self, *a, **kw)
        torch.cuda.synchronize()
        proxy.log.append(("heavy_keys_sample", round((time.perf_counter() - t0) * 1e3, 3)))
        return r
    Frame._heavy_keys = timed_heavy
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sa.timer_start(0)
        res = frame.groupby("k", spec)
        sa.timer_stop(0)
        k_ms = sa.timer_kernels_ms(0)
        ms = (time.perf_counter() - t0) * 1e3
    finally:
        Frame._heavy_keys = heavy
    info = dict(getattr(frame, "last_groupby_info", None) or {})
    return {"ms": round(ms, 3), "kernel_ms": round(k_ms, 3), "groups": int(len(res["k"])), "phases": list(proxy.log),
            "info": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in info.items()}}


g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
free0 = torch.cuda.mem_get_info()[0]
out = []
for flavour in ("dense", "scattered"):
    keys = k if flavour == "dense" else (k * 2654435761) % (1 << 40)
    torch.cuda.synchronize()
    for r in range(rounds):
        proxy = Timed()
        # bench.py's sequence: prime over clones (first round: the process's first call of this kind), empty_cache, first call over fresh columns, warm calls
        mode = ["prime+empty_cache", "prime", "fresh clones", "fresh clones+empty_cache"][r % 4] if r else "prime+empty_cache"
        rec = {"flavour": flavour, "round": r, "mode": mode, "free_gb_before": round(torch.cuda.mem_get_info()[0] / 2**30, 1)}
        if mode.startswith("prime"):
            p = Timed()
            rec["prime"] = call(Frame(dict(k=keys.clone(), v=v.clone()), superagg=p), spec, p)
            if mode.endswith("empty_cache"):
                t0 = time.perf_counter()
                torch.cuda.empty_cache()
                rec["empty_cache_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            df = Frame(dict(k=keys, v=v), superagg=proxy) if r == 0 else Frame(dict(k=keys.clone(), v=v.clone()), superagg=proxy)
        else:
            df = Frame(dict(k=keys.clone(), v=v.clone()), superagg=proxy)
            if mode.endswith("empty_cache"):
                t0 = time.perf_counter()
                torch.cuda.empty_cache()
                rec["empty_cache_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        rec["first"] = call(df, spec, proxy)
        rec["warm"] = [call(df, spec, proxy) for _ in range(2)]
        rec["free_gb_after"] = round(torch.cuda.mem_get_info()[0] / 2**30, 1)
        out.append(rec)
        print(json.dumps({"flavour": flavour, "round": r, "mode": mode, "prime_ms": rec.get("prime", {}).get("ms"), "empty_cache_ms": rec.get("empty_cache_ms"),
                          "first_ms": rec["first"]["ms"], "first_kernel_ms": rec["first"]["kernel_ms"], "first_phases": rec["first"]["phases"],
                          "first_retries": rec["first"]["info"].get("retries"), "warm_ms": [w["ms"] for w in rec["warm"]], "free_gb": [rec["free_gb_before"], rec["free_gb_after"]]}), flush=True)
        del df
import os
rep = os.environ.get("VAEX_AMD_REPORT_DIR")
if rep:
    json.dump(out, open(os.path.join(rep, "first_call.json"), "w"), indent=1)
firsts = {f: sorted(r["first"]["ms"] for r in out if r["flavour"] == f) for f in ("dense", "scattered")}
print(json.dumps({f: {"first_ms_min_median_max": [v_[0], v_[len(v_) // 2], v_[-1]]} for f, v_ in firsts.items()}))
