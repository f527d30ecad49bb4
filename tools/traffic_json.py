#!/usr/bin/env python3
"""profiles/r<NN>_traffic.json from the PMC summaries of a profiling call (the bench pass's: pmc_bench_traffic.txt or pmc_<n>.txt; the configs': configs_pmc_traffic.txt):
HBM bytes per row = sum over the kernels of a pass of (FETCH_SIZE x 2 + WRITE_SIZE) KiB per dispatch x dispatches per pass / rows.
FETCH_SIZE x 2: gfx950 counts 128-byte requests as 64 B (MI355X_MICROARCH.md, HBM section; the check printed with every config:
the dominant kernel's read bytes per row against the columns it must read).
Usage: python tools/traffic_json.py <bench pmc summary> <configs pmc summary> [round label] > profiles/r05_traffic.json"""
import json, re, sys, os

def parse(path):
    out, cur, sect = {}, None, "bench"
    for ln in open(path):
        m = re.match(r"=== (\S+)", ln)
        if m:
            sect = m.group(1)
            continue
        if not ln.startswith(" "):
            cur = ln.strip()
            out.setdefault(sect, {}).setdefault(cur, {})
            continue
        m = re.match(r"\s+(\S+)\s+([\d.]+) per dispatch\s+\((\d+) dispatches\)", ln)
        if m and cur:
            out[sect][cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return out

bench = parse(sys.argv[1]).get("bench", {})
cfgs = parse(sys.argv[2]) if len(sys.argv) > 2 and os.path.exists(sys.argv[2]) else {}
label = sys.argv[3] if len(sys.argv) > 3 else "6"
ROWS = 1_000_000_000
# kernels of one pass: (name prefix, dispatches per pass)
SPEC = {
    "bench": ([("part_scatter_wv", 1), ("part_reduce_", 1), ("part_merge", 1), ("part_hot_merge", 1)], 24),
    "count2d": ([("part_scatter_wv", 1), ("part_reduce_", 1), ("part_merge", 1), ("part_hot_merge", 1)], 16),   # (round 5: through the hot box; rounds 3-4: count_lds_f64 + fold_kernel)
    "c2": ([("part_scatter_wv", 4), ("part_reduce_fast", 4), ("part_merge", 1)], 25),
    "c2e": ([("part_scatter_wv", 4), ("part_reduce_fast", 4), ("part_merge", 1)], 32),
    "c3d": ([("gb_scatter", 1), ("gb_reduce", 1)], 16),   # (round 6: the dense range through the fused pass with a direct table; rounds 3-5: part_scatter_f64 x2 + part_reduce_fast x2 + part_merge)
    "c3s": ([("gb_scatter", 1), ("gb_reduce", 1)], 16),
}

def per_row(kernels, spec):
    total, parts = 0.0, {}
    for prefix, n in spec:
        # the instantiation of this prefix with the most dispatches (a first call's timed trial launches the other form once)
        cands = [(v.get("FETCH_SIZE", (0, 0))[1], k, v) for k, v in kernels.items() if k.startswith(prefix) and "FETCH_SIZE" in v]
        if not cands:
            continue
        _, k, v = max(cands)
        f, w = v["FETCH_SIZE"][0], v.get("WRITE_SIZE", (0.0, 0))[0]
        b = (2.0 * f + w) * 1024.0 * n / ROWS
        parts[k + f" x{n}"] = {"read_B_per_row": round(2.0 * f * 1024.0 * n / ROWS, 3), "written_B_per_row": round(w * 1024.0 * n / ROWS, 3)}
        total += b
    return round(total, 2), parts

out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of bench.py and of tools/r03_config_one.py <config> 1e9 2 on the final tree of round " + label + " (tools/gpu_call.sh steps `pmc` and `cfgprof`; summaries: profiles/r0" + label + "_pmc_bench.txt, r0" + label + "_configs_pmc_traffic.txt)",
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section)", "rows_per_pass": ROWS}
t, parts = per_row(bench, SPEC["bench"][0])
out["hbm_bytes_per_row"] = t
out["algorithmic_bytes_per_row"] = SPEC["bench"][1]
out["breakdown"] = parts
out["configs"] = {}
for name, (spec, alg) in SPEC.items():
    if name == "bench" or name not in cfgs:
        continue
    t, parts = per_row(cfgs[name], spec)
    out["configs"][name] = {"hbm_bytes_per_row": t, "algorithmic_bytes_per_row": alg, "ratio": round(t / alg, 2), "breakdown": parts}
json.dump(out, sys.stdout, indent=1)
print()
