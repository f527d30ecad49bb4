"""Round 6 (late): the key-packing pass of a multi-key groupby (vxh_pack_keys) — pack_keys_n (loads of a round issued up front) against the
row-at-a-time kernel, 1e9 rows, best of 10.  (The comparison in profiles/r06_pack_keys.txt was taken with a build that could still be told to launch
the old kernel for 1-4 keys — VAEX_HIP_PACK_KEYS_GENERIC — a switch the product library no longer has; five keys and more still take the old kernel.)
    python tools/r09_pack_keys.py [rows]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, vaex_amd
sa = vaex_amd.superagg
sa.warmup()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
DT = {"int64": 2, "int32": 3, "int16": 4, "int8": 5, "uint8": 9}
TD = {"int64": torch.int64, "int32": torch.int32, "int16": torch.int16, "int8": torch.int8, "uint8": torch.uint8}
which = "pack_keys_n (1-4 keys)"
for kinds in (["int64", "int32"], ["int64", "int64"], ["int32", "int32", "int16"], ["int32", "int16", "int8", "uint8"], ["int64"]):
    cols = [torch.randint(0, 100, (n,), device="cuda", dtype=torch.int32).to(TD[k]) for k in kinds]
    mins = [0] * len(kinds)
    mults = [100 ** i for i in range(len(kinds))]
    best = 1e9
    for _ in range(10):
        sa.slot_wait(0); torch.cuda.synchronize(); t0 = time.perf_counter(); out = sa.pack_keys(cols, [DT[k] for k in kinds], mins, mults); sa.slot_wait(0); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
        del out
    nbytes = sum(c.element_size() for c in cols) + 8
    print(f"{which:22s} {'+'.join(kinds):28s} {best:7.2f} ms  ({nbytes} B/row: {nbytes * n / best / 1e9:5.2f} TB/s)", flush=True)
    del cols
