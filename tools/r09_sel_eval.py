"""Round 6 (late): the mask pass of a selection `column <op> constant` — sel_eval_vec (whole quads, loads up front; columns aligned for it) against the
generic sel_eval (the same values one element off that alignment), per column dtype; wall clock around Selection.evaluate + slot_wait, best of 10.
    python tools/r09_sel_eval.py [rows]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, vaex_amd
sa = vaex_amd.superagg
sa.warmup()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
out = torch.empty((n + 3) & ~3, dtype=torch.uint8, device="cuda")
kinds = [("float64", 0, torch.float64), ("float32", 1, torch.float32), ("int64", 2, torch.int64), ("int32", 3, torch.int32), ("int16", 4, torch.int16), ("uint8", 9, torch.uint8)]
for name, code, dt in kinds:
    base = (torch.randn(n + 1, device="cuda") * 3).to(dt) if dt.is_floating_point else torch.randint(0, 100, (n + 1,), device="cuda", dtype=torch.int32).to(dt)
    line = f"{name:8s}"
    masks = []
    for label, col in (("quads", base[:n]), ("generic", base[1:])):
        res = []
        for two in (False, True):
            terms = [(0, 2, 1)] if not two else [(0, 2, 1), (0, 0, 50)]
            sel = sa.Selection(1, [code], terms, 0b10 if not two else 0b1000)
            sel.set_data(0, 0, col)
            best = 1e9
            for _ in range(10):
                sa.slot_wait(0); t0 = time.perf_counter(); sel.evaluate(0, n, out); sa.slot_wait(0); best = min(best, (time.perf_counter() - t0) * 1e3)
            res.append(best)
        isz = base.element_size()
        line += f"   {label}: 1 term {res[0]:6.2f} ms ({(isz + 1) * n / res[0] / 1e9:5.2f} TB/s)  2 terms {res[1]:6.2f} ms"
    print(line, flush=True)
    del base
