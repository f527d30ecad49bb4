"""Round 6: run bench.py's configs section N times and print every config line's first call that took more than 2.5x its warm call, with the line's own
account of it (library calls, Frame steps, allocator counters).  python tools/r06_catch_first_call.py [N=8]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
slow = 0
for i in range(n):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extra", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, cwd=ROOT, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        print("run", i, "no line", p.stderr[-500:])
        continue
    d = json.loads(lines[-1])
    row = []
    for c in d.get("configs", []):
        ratio = c["ms_first_call"] / c["ms"]
        row.append("%s %.1f/%.1f" % (c["config"], c["ms_first_call"], c["ms"]))
        if ratio > 2.5:
            slow += 1
            print("SLOW FIRST CALL run %d %s: %s" % (i, c["config"], json.dumps(c.get("first_call"))), flush=True)
    print("run", i, " | ".join(row), flush=True)
print("slow first calls:", slow, "of", n, "runs")
