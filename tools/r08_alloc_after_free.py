"""Round 6 (late): what a large hipMalloc costs right after a large hipFree.  bench.py's `ms_first_call_in_process` of the dense groupby is 14-16 ms on some runs and
0.5-0.9 s on others; the line's block-pool clock puts the difference in ONE hipMalloc of 24.5 GB (r08n1: 664 ms of 679), and a 20 GB torch allocation right behind a
torch.cuda.empty_cache() of tens of GB took 2.3-3.3 s on the same boxes.  Hypothesis: the driver scrubs VRAM it got back, and an allocation that follows waits for it.
Rounds: allocate + touch G GB, then a 20 GB allocation (a) with nothing freed before, (b) right behind the release of the G GB, (c) behind the release and a pause."""
import sys
import time

import torch

gbs = [float(a) for a in sys.argv[1:]] or [8.0, 32.0, 64.0]


def alloc_ms(nbytes):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t = torch.empty(int(nbytes), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    return dt, t


torch.zeros(1, device="cuda")
for g in gbs:
    for mode in ("nothing freed before", "right behind the release", "behind the release and a 6 s pause", "right behind the release"):
        big = torch.empty(int(g * 2**30), dtype=torch.uint8, device="cuda")
        big.fill_(1)
        torch.cuda.synchronize()
        t_free = None
        if mode != "nothing freed before":
            del big
            t0 = time.perf_counter()
            torch.cuda.empty_cache()
            torch.cuda.synchronize()
            t_free = round((time.perf_counter() - t0) * 1e3, 2)
            if "pause" in mode:
                time.sleep(6.0)
        dt, blk = alloc_ms(20 * 2**30)
        dt2, blk2 = alloc_ms(20 * 2**30)   # (a second one straight after)
        print({"held_or_freed_gb": g, "mode": mode, "release_ms": t_free, "alloc20GB_ms": round(dt, 2), "second_alloc20GB_ms": round(dt2, 2)}, flush=True)
        del blk, blk2
        if mode == "nothing freed before":
            del big
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        time.sleep(8.0)   # (whatever that release started is over before the next round)
print("DONE")
