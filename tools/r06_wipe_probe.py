"""Round 6 hypothesis for the 460 ms first call of bench.py's configs[3]' (seen on the driver's round-5 box and on one of five builder boxes): the timed call
directly follows torch.cuda.empty_cache() of the 16 GB of primed clones.  The kernel driver wipes VRAM on release; work that needs memory from a region still being
wiped — or simply queued behind the wipe — waits.  Probe: free G gigabytes through empty_cache(), then time (a) a kernel over memory that already exists,
(b) a fresh 64 MB torch allocation + fill, (c) a fresh 2 GB allocation + fill; rounds with and without the free.
    python tools/r06_wipe_probe.py [gb=16] [rounds=24]"""
import sys
import time

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 24
keep = torch.zeros(1 << 26, dtype=torch.float64, device="cuda")   # 512 MB that stays
torch.cuda.synchronize()


def ms(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 2)


def fresh(nbytes):
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    t.fill_(1)
    return t


worst = {}
for r in range(rounds):
    free_it = r % 3 != 2
    big = [torch.empty(int(gb * (1 << 30) // 2), dtype=torch.uint8, device="cuda") for _ in range(2)]
    for b in big:
        b.fill_(3)
    torch.cuda.synchronize()
    del big, b
    t_free = None
    if free_it:
        t0 = time.perf_counter()
        torch.cuda.empty_cache()
        t_free = round((time.perf_counter() - t0) * 1e3, 2)
    a = ms(lambda: keep.add_(1.0))
    hold = []
    b_ = ms(lambda: hold.append(fresh(64 << 20)))
    c = ms(lambda: hold.append(fresh(2 << 30)))
    d = ms(lambda: keep.add_(1.0))
    del hold
    print({"round": r, "freed_gb": gb if free_it else 0, "empty_cache_ms": t_free, "kernel_existing_ms": a, "alloc64MB_fill_ms": b_, "alloc2GB_fill_ms": c, "kernel_after_ms": d}, flush=True)
    for k_, v_ in (("kernel_existing_ms", a), ("alloc64MB_fill_ms", b_), ("alloc2GB_fill_ms", c)):
        key = (k_, free_it)
        worst[key] = max(worst.get(key, 0), v_)
    if not free_it:
        torch.cuda.empty_cache()
        time.sleep(1.0)
print({"worst_ms": {"%s after_free=%s" % k_: v_ for k_, v_ in worst.items()}})
