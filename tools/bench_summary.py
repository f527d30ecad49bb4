#!/usr/bin/env python3
"""One-screen summary of a bench.py JSON line; with a kernel-stats file of the SAME process: sum of its pass kernels vs kernel_ms."""
import json
import re
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("headline %.4g rows/s  ms/step %.3f  kernel_ms %.3f  frac %.4f  cold %s  uniform %s  sustained %s  %s" % (
    d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], ("%.3f" % r["frac_cold"]) if "frac_cold" in r else "-",
    ("%.3f" % r["frac_uniform"]) if "frac_uniform" in r else "-", ("%.3f" % d["sustained"]["frac"]) if "sustained" in d else "-", d["config"]["kernel"]))
if "allreduce_world1_ms" in d or "allreduce_world1" in d:
    print("  allreduce_world1_ms", d.get("allreduce_world1_ms"), d.get("allreduce_world1"))
for c in d.get("configs") or []:
    print("  %-12s %.3g rows/s kernel_ms %.3f frac %.3f first_call_ms %s ok %s %s" % (c.get("config"), c.get("rows_per_s", 0), c.get("kernel_ms", 0), c.get("roofline", {}).get("frac", 0),
          c.get("ms_first_call"), (c.get("parity_on_sample") or {}).get("ok"), c.get("groupby_kernels_ms", "")))
    pf = c.get("first_call_in_process") or {}
    if c.get("ms_first_call_in_process") is not None:
        print("      process-first %.1f ms  pool hipMalloc %s us over %s bytes  calls %s  %s" % (c["ms_first_call_in_process"], (pf.get("pool") or {}).get("pool_malloc_us"), (pf.get("pool") or {}).get("pool_malloc_bytes"),
              pf.get("library_calls_ms"), c.get("large_alloc_probe", "")))
cb = d.get("cpu_baseline") or {}
if cb:
    print("  cpu %.4g rows/s on %s cores (%s) parity %s  vaex: %s" % (cb.get("value", 0), cb.get("cores"), cb.get("kind"), cb.get("parity_on_sample"), ({k: cb["through_vaex"].get(k) for k in ("value", "threads", "runs", "error")} if cb.get("through_vaex") else None)))
if len(sys.argv) > 2:
    tot, steps = 0.0, d["steps"] + d["warmup"]
    for line in open(sys.argv[2]):
        m = re.match(r"(part_\S+|count_lds\S*|bin_kernel\S*|fold_kernel\S*).*?\s(\d+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s*$", line)
        if m:
            tot += float(m.group(4))
    print("  sum of the pass kernels: %.3f ms per step over %d steps (kernel_ms of this same process: %.3f; ratio %.3f)" % (tot / steps, steps, r["kernel_ms"], tot / steps / r["kernel_ms"]))
