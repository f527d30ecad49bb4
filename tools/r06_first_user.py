"""Round 6 diagnosis (VERDICT r5 weak #1b): is a process that is the FIRST GPU user of a fresh box — or simply a fresh process — able to die in
vaex_amd.install()?  N fresh subprocesses, each: import vaex (the reference's package from oracle/_ref), vaex_amd.install(), one binned count and one
groupby through the HIP classes; rc, seconds, and the whole stderr of every process that did not print its OK line are kept.
    python tools/r06_first_user.py [N=60] [parallel=1]"""
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VAEXPY = os.path.join(ROOT, "oracle", "_ref", "vaexpy")
PKG = VAEXPY if os.path.isdir(os.path.join(VAEXPY, "vaex")) else os.path.join(ROOT, "oracle", "_ref", "overlay")
FAKE = os.path.join(ROOT, "oracle", "fake")

CHILD = r'''
import sys, time
t0 = time.perf_counter()
import numpy as np
import vaex, vaex_amd
t1 = time.perf_counter()
n_dev = vaex_amd.superagg.device_count()
t2 = time.perf_counter()
assert n_dev > 0, "device_count() == 0"
vaex_amd.install()
t3 = time.perf_counter()
rng = np.random.default_rng(1)
n = 300_000
df = vaex.from_arrays(x=rng.normal(0, 1, n), y=rng.normal(0, 1, n), k=rng.integers(0, 50, n))
c = df.count(binby=["x", "y"], limits=[[-4, 4], [-4, 4]], shape=64)
g = df.groupby("k", agg={"c": "count", "m": vaex.agg.mean("x")})
t4 = time.perf_counter()
assert int(c.sum()) > 0.99 * n and len(g) == 50
print("OK import %.2f device_count %.3f install %.3f calls %.3f hip_parts %d" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, vaex_amd.task_stats.get("hip", 0)))
'''


def one(i):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, FAKE, ROOT]), VAEX_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600, cwd="/tmp")
    dt = time.perf_counter() - t0
    ok = p.returncode == 0 and "OK import" in p.stdout
    return {"i": i, "rc": p.returncode, "s": round(dt, 2), "ok": ok, "line": p.stdout.strip().splitlines()[-1] if p.stdout.strip() else "",
            "stderr": "" if ok else p.stderr[-6000:], "stdout": "" if ok else p.stdout[-2000:]}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    par = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = []
    first = one(0)     # alone: on a fresh box this process is the first user of the GPU
    out.append(first)
    print("first:", json.dumps(first)[:600], flush=True)
    with ThreadPoolExecutor(par) as pool:
        for r in pool.map(one, range(1, n)):
            out.append(r)
            if not r["ok"]:
                print("FAILED:", json.dumps(r)[:3000], flush=True)
    bad = [r for r in out if not r["ok"]]
    secs = sorted(r["s"] for r in out)
    print(json.dumps({"runs": len(out), "parallel": par, "failed": len(bad), "seconds_min_median_max": [secs[0], secs[len(secs) // 2], secs[-1]], "first": out[0]["line"], "last": out[-1]["line"]}))
    rep = os.environ.get("VAEX_AMD_REPORT_DIR")
    if rep:
        json.dump(out, open(os.path.join(rep, "first_user_%d.json" % par), "w"), indent=1)


if __name__ == "__main__":
    main()
