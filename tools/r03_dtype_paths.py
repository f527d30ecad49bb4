#!/usr/bin/env python3
"""Round 3: the bench pass (2-D 256x256, count + sum + count(v), 1e9 N(0,1) rows) with other value dtypes — which kernels run, how fast.
Usage: python tools/r03_dtype_paths.py [rows] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
vals = {
    "float64": torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3,
    "int64": torch.randint(-1000, 1000, (rows,), dtype=torch.int64, device="cuda", generator=g),
    "int32": torch.randint(-1000, 1000, (rows,), dtype=torch.int32, device="cuda", generator=g),
    "float32": torch.randn(rows, dtype=torch.float32, device="cuda", generator=g),
}
torch.cuda.synchronize()
x32, y32 = x.to(torch.float32), y.to(torch.float32)
for name, v in list(vals.items()) + [("f32bin+float64", vals["float64"]), ("f32bin+int64", vals["int64"]), ("f32bin+float32", vals["float32"])]:
    f32bin = name.startswith("f32bin+")
    name = name.replace("f32bin+", "")
    B = sa.BinnerScalar_float32 if f32bin else sa.BinnerScalar_float64
    bx = B(1, "x", -4.0, 4.0, 256); by = B(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    aggs = [sa.AggCount_int64(grid, 1, 1), getattr(sa, "AggSum_" + name)(grid, 1, 1), getattr(sa, "AggCount_" + name)(grid, 1, 1)]
    bx.set_data(0, x32 if f32bin else x); by.set_data(0, y32 if f32bin else y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
    best = 1e9
    for r in range(reps + 1):
        for a in aggs:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, aggs, rows)
        ms = sa.timer_stop(0)
        if r:
            best = min(best, ms)
    total = int(np.array(aggs[0].get_result()).sum())
    assert total == rows
    bytes_per_row = (8 if f32bin else 16) + v.element_size()
    print(f"binners {'float32' if f32bin else 'float64'} value {name:<8} {best:8.3f} ms {rows/best/1e6:7.1f} Grows/s  {rows*bytes_per_row/best/1e6/8000:6.3f} of 8 TB/s on {bytes_per_row} B/row   {sa.last_kernel(0)}", flush=True)

# round 4: integer BINNER columns (converted on load by part_scatter_wv: PartArgs::bin_ct 2 / 3), 256 x 256 cells: 8 slabs
for bname, tdt in (("int64", torch.int64), ("int32", torch.int32)):
    xi = torch.randint(0, 1024, (rows,), dtype=torch.int64, device="cuda", generator=g).to(tdt)
    yi = torch.randint(0, 1024, (rows,), dtype=torch.int64, device="cuda", generator=g).to(tdt)
    B = getattr(sa, "BinnerScalar_" + bname)
    for knob in (1, 0):   # 0: what ran before (the generic pair)
        sa.config_set("wv", 6 if knob else 0); sa.config_set("blk", 1 if knob else 0)
        bx = B(1, "x", 0.0, 1024.0, 256); by = B(1, "y", 0.0, 1024.0, 256)
        grid = sa.Grid([bx, by])
        v = vals["float64"]
        aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
        bx.set_data(0, xi); by.set_data(0, yi); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
        best = 1e9
        for r in range(reps + 1):
            for a in aggs:
                a.reset()
            sa.timer_start(0)
            grid.bin(0, aggs, rows)
            ms = sa.timer_stop(0)
            if r:
                best = min(best, ms)
        assert int(np.array(aggs[0].get_result()).sum()) == rows
        bpr = 2 * xi.element_size() + 8
        print(f"binners {bname:<7} value float64 (256x256 uniform) {best:8.3f} ms {rows/best/1e6:7.1f} Grows/s  {rows*bpr/best/1e6/8000:6.3f} of 8 TB/s on {bpr} B/row   {sa.last_kernel(0)}", flush=True)
    sa.config_set("wv", 6); sa.config_set("blk", 1)
    del xi, yi

# round 5: binner columns of the dtypes the fast kernels do not read — converted to float64 by a pass of their own ("convert_binners") —
# on the bench shape with N(0,1)-like data (the values of x, y scaled into the integer type), against the generic pair (convert_binners=0)
for bname, tdt, scale in (("int16", torch.int16, 1000.0), ("int8", torch.int8, 30.0), ("uint8", torch.uint8, 30.0), ("uint16", torch.int16, 1000.0)):
    off = 128.0 if bname == "uint8" else (32768.0 / 2 if bname == "uint16" else 0.0)
    xi = (x * scale + off).clamp(-32768 if bname != "uint8" else 0, 32767 if "16" in bname else (255 if bname == "uint8" else 127)).to(tdt if bname != "uint8" else torch.uint8)
    yi = (y * scale + off).clamp(-32768 if bname != "uint8" else 0, 32767 if "16" in bname else (255 if bname == "uint8" else 127)).to(tdt if bname != "uint8" else torch.uint8)
    B = getattr(sa, "BinnerScalar_" + bname)
    lo, hi = off - 4 * scale, off + 4 * scale
    for conv in (1 << 22, 0):
        sa.config_set("convert_binners", conv)
        bx = B(1, "x", lo, hi, 256); by = B(1, "y", lo, hi, 256)
        grid = sa.Grid([bx, by])
        v = vals["float64"]
        aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
        bx.set_data(0, xi); by.set_data(0, yi); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
        best = 1e9
        for r in range(reps + 1):
            for a in aggs:
                a.reset()
            sa.timer_start(0)
            grid.bin(0, aggs, rows)
            ms = sa.timer_stop(0)
            if r:
                best = min(best, ms)
        assert int(np.array(aggs[0].get_result()).sum()) == rows
        bpr = 2 * xi.element_size() + 8
        print(f"binners {bname:<7} value float64 {'converted to float64 first' if conv else 'generic kernels         '} {best:8.3f} ms {rows/best/1e6:7.1f} Grows/s  {rows*bpr/best/1e6/8000:6.3f} of 8 TB/s on {bpr} B/row   {sa.last_kernel(0)}", flush=True)
    sa.config_set("convert_binners", 1 << 22)
    del xi, yi

# round 6: VALUE columns of the dtypes the typed paths do not load — converted to int64 by a pass of their own — float64 binners, bench shape,
# against the generic pair (convert_binners=0)
for vname, tdt in (("int16", torch.int16), ("int8", torch.int8), ("uint8", torch.uint8)):
    vv = torch.randint(0, 100, (rows,), dtype=torch.int64, device="cuda", generator=g).to(tdt)
    for conv in (1 << 22, 0):
        sa.config_set("convert_binners", conv)
        bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
        grid = sa.Grid([bx, by])
        aggs = [sa.AggCount_int64(grid, 1, 1), getattr(sa, "AggSum_" + vname)(grid, 1, 1), getattr(sa, "AggCount_" + vname)(grid, 1, 1)]
        bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, vv, 0); aggs[2].set_data(0, vv, 0)
        best = 1e9
        for r in range(reps + 1):
            for a in aggs:
                a.reset()
            sa.timer_start(0)
            grid.bin(0, aggs, rows)
            ms = sa.timer_stop(0)
            if r:
                best = min(best, ms)
        assert int(np.array(aggs[0].get_result()).sum()) == rows
        bpr = 16 + vv.element_size()
        print(f"binners float64 value {vname:<7} {'converted to int64 first' if conv else 'generic kernels        '} {best:8.3f} ms {rows/best/1e6:7.1f} Grows/s  {rows*bpr/best/1e6/8000:6.3f} of 8 TB/s on {bpr} B/row   {sa.last_kernel(0)}", flush=True)
    sa.config_set("convert_binners", 1 << 22)
    del vv
