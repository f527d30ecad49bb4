"""Round 6 (late): do the runtime's allocator stalls (profiles/r06_alloc_probe.txt) depend on the SIZE of one allocation?  Rounds of 20 GB taken as 1 x 20 GB, 10 x 2 GB and
80 x 256 MB through hipMalloc (torch's allocator with its cache emptied), every round timed; a stall is a round over 100 ms."""
import sys
import time

import torch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def take(n, each):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    blocks = [torch.empty(each, dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    del blocks
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    return dt


torch.zeros(1, device="cuda")
forms = {"1 x 20 GB": (1, 20 << 30), "10 x 2 GB": (10, 2 << 30), "80 x 256 MB": (80, 256 << 20)}
times = {k: [] for k in forms}
for r in range(rounds):
    for k, (n, each) in forms.items():
        times[k].append(round(take(n, each), 1))
        time.sleep(0.2)
for k, v in times.items():
    print(k, "| stalls (> 100 ms):", sum(1 for q in v if q > 100), "of", len(v), "| median ms", sorted(v)[len(v) // 2], "| the stalls:", [q for q in v if q > 100], flush=True)
print("DONE")
