"""Round 6 (late): which state of the HOST side of a process makes a large hipMalloc slow?  In bench.py a 20 GB allocation cost 0.4 ms in front of the CPU baseline and
2.3-3.7 s behind it (profiles/r06_alloc_probe.txt).  Here: the same probe after (a) nothing, (b) a pool of 256 threads that ran and was shut down, (c) gigabytes of host
arrays allocated, touched and dropped by the main thread, (d) both at once (what the baseline does), each followed by probes every half second."""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def probe(gb=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = torch.empty(gb << 30, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    del t
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    return round(dt, 2)


def churn(seconds, nbytes=140 << 20):
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        a = np.zeros(nbytes // 8)      # (mmap'ed by glibc at this size, touched, unmapped when dropped)
        a += 1.0
        del a
        n += 1
    return n


def threads(nthreads, seconds, with_churn):
    def work(_):
        return churn(seconds, 16 << 20) if with_churn else time.sleep(seconds)
    with ThreadPoolExecutor(nthreads) as pool:
        list(pool.map(work, range(nthreads)))


x = torch.zeros(3 << 27, dtype=torch.float64, device="cuda")   # 3 GB that stay
host = x.cpu().numpy()
print("fresh process:", [probe() for _ in range(3)], flush=True)
for label, fn in (("256 idle threads, pool shut down", lambda: threads(256, 2.0, False)),
                  ("host arrays made and dropped by the main thread (5 s)", lambda: churn(5.0)),
                  ("256 threads making and dropping host arrays (3 s)", lambda: threads(256, 3.0, True)),
                  ("32 threads making and dropping host arrays (3 s)", lambda: threads(32, 3.0, True))):
    fn()
    out = []
    for i in range(8):
        out.append(probe())
        time.sleep(0.5)
    print(label, "->", out, flush=True)
print("DONE")
