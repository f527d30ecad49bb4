"""Round 6 (late): a filter over the groupby's own value column — terms evaluated inside gb_scatter (vxh_groupby_run_selected) against the
keep-mask road (one sel_eval pass writing a byte per row + gb_scatter reading it back).  1e9 rows x 1e6 int64 keys, `v > 3` keeps half.
    python tools/r08_groupby_pred.py [rows]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, vaex_amd
from vaex_amd import binned
sa = vaex_amd.superagg
sa.warmup()
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
for flavour in ("dense", "scattered"):
    k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
    if flavour == "scattered":
        k = (k * 2654435761) % (1 << 40)
    torch.cuda.synchronize()
    kr = (int(k.min()), int(k.max()))
    f = binned.Frame(dict(k=k, v=v), superagg=sa)
    pred = f._groupby_pred_terms("v > 3", ["v"])

    def mask():
        f.__dict__.pop("_device_masks", None)
        return f._mask_array("v > 3")

    def timed(fn, reps=5):
        out = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
        return r, out
    m, t_mask = timed(mask)
    r_keep, t_keep = timed(lambda: sa.groupby_run(k, [v], 2, keep=m, key_range=kr))
    r_pred, t_pred = timed(lambda: sa.groupby_run(k, [v], 2, key_range=kr, pred=pred))
    r_all, t_all = timed(lambda: sa.groupby_run(k, [v], 2, key_range=kr))
    col = lambda r, c: np.asarray(r.column(c, 0))
    same = all(np.array_equal(col(r_keep, c), col(r_pred, c)) for c in (sa.GB_KEYS, sa.GB_ROWS, sa.GB_COUNT)) and \
        all(np.allclose(col(r_keep, c), col(r_pred, c), rtol=1e-11, atol=0) for c in (sa.GB_SUM, sa.GB_SUM2))   # (the sums are LDS atomics: their order is the pass's own)
    fmt = lambda t: f"{min(t):8.2f} ms (median {float(np.median(t)):.2f})"
    print(f"{flavour:9s} {rows:.3g} rows, {len(r_pred)} groups of the kept half; keys / rows / counts identical, sums to 1e-11: {same}")
    print(f"   no filter                        {fmt(t_all)}")
    print(f"   keep-mask: sel_eval pass         {fmt(t_mask)}")
    print(f"   keep-mask: groupby over the mask {fmt(t_keep)}   first call = mask + groupby = {min(t_mask) + min(t_keep):.2f} ms")
    print(f"   terms inside gb_scatter          {fmt(t_pred)}   x{(min(t_mask) + min(t_keep)) / min(t_pred):.2f} first call, x{min(t_keep) / min(t_pred):.2f} with the mask cached", flush=True)
    del f, m, r_keep, r_pred, r_all, k
