"""The chunk feeder and the device column cache (include/vaex_hip.h "chunk feeder and device column cache"):
host chunks handed to Grid.bin the way vaex's executor does (vaex/execution.py:432-435 — one chunk per call, the slot of the
calling thread, arrays that may be overwritten as soon as the call returns) must give the same grids as device-resident
columns, through the pinned ring, through plain copies, from registered (cached / page-locked) columns, with evictions, and
from several threads at once."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sa = pytest.importorskip("vaex_amd.superagg")


@pytest.fixture(autouse=True)
def _restore_knobs():
    keep = {k: sa.config_get(k) for k in ("feeder", "cache_bytes", "stage_bytes")}
    yield
    for k, v in keep.items():
        sa.config_set(k, v)
    import vaex_amd
    vaex_amd.uncache_columns()
    sa.cache_clear()


def _expected(x, v, lo, hi, bins, sel=None):
    ok = ~np.isnan(x) & (x >= lo) & (x < hi)
    if sel is not None:
        ok &= sel.astype(bool)
    idx = ((x[ok] - lo) / (hi - lo) * bins).astype(np.int64)
    c = np.bincount(idx, minlength=bins)
    s = np.bincount(idx, weights=v[ok], minlength=bins)
    return c, s


def _stream(x, v, chunk, slots=1, sel=None, overwrite=False, bins=64):
    """bin x (count, sum v) chunk by chunk from host memory; overwrite: hand over a scratch buffer that is clobbered right after
    Grid.bin returns (an evaluated expression's buffer in vaex)"""
    b = sa.BinnerScalar_float64(slots, "x", -4.0, 4.0, bins)
    g = sa.Grid([b])
    ac = sa.AggCount_float64(g, slots, slots)
    asum = sa.AggSum_float64(g, slots, slots)
    n = len(x)
    starts = list(range(0, n, chunk))
    lock = threading.Lock()

    def run(slot):
        bx, bv = np.empty(chunk), np.empty(chunk)
        bm = np.empty(chunk, dtype="u1")
        while True:
            with lock:
                if not starts:
                    return
                i1 = starts.pop(0)
            i2 = min(n, i1 + chunk)
            if overwrite:
                bx[: i2 - i1] = x[i1:i2]; bv[: i2 - i1] = v[i1:i2]
                cx, cv = bx[: i2 - i1], bv[: i2 - i1]
            else:
                cx, cv = x[i1:i2], v[i1:i2]
            b.set_data(slot, cx)
            ac.set_data(slot, cx, 0)
            asum.set_data(slot, cv, 0)
            if sel is not None:
                if overwrite:
                    bm[: i2 - i1] = sel[i1:i2]
                    cm = bm[: i2 - i1]
                else:
                    cm = sel[i1:i2]
                ac.set_data_mask(slot, cm); asum.set_data_mask(slot, cm)
            g.bin(slot, [ac, asum], i2 - i1)
            if overwrite:
                bx[:] = np.nan; bv[:] = 1e300; bm[:] = 0

    threads = [threading.Thread(target=run, args=(s,)) for s in range(slots)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return np.asarray(ac.get_result())[2:-1], np.asarray(asum.get_result())[2:-1]


def _data(n, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1.5, n)
    x[::97] = np.nan
    v = rng.integers(-1000, 1000, n).astype("f8")  # integer-valued: sums are exact in any order
    sel = (rng.random(n) < 0.6).astype("u1")
    return x, v, sel


@pytest.mark.parametrize("feeder", [1, 2, 0])
@pytest.mark.parametrize("slots", [1, 4])
def test_host_chunks_through_the_ring(feeder, slots):
    sa.config_set("feeder", feeder)
    sa.config_set("stage_bytes", 1 << 20)  # (arenas grow on the larger chunks below)
    x, v, sel = _data(1_300_003)
    want = _expected(x, v, -4, 4, 64, sel)
    for chunk in (10_007, 65_536, 400_000):  # many more calls than ring entries; ragged last chunk
        c, s = _stream(x, v, chunk, slots=slots, sel=sel, overwrite=True)
        assert np.array_equal(c, want[0]) and np.array_equal(s, want[1]), (feeder, slots, chunk)


def test_registered_columns_are_served_from_hbm_on_the_second_pass():
    import vaex_amd
    x, v, sel = _data(2_000_000, 1)
    want = _expected(x, v, -4, 4, 64)
    for pin in (True, False):
        assert vaex_amd.cache_columns({"x": x, "v": v}, pin=pin) == x.nbytes + v.nbytes
        st0 = sa.cache_stats()
        assert st0["ranges"] == 2
        c, s = _stream(x, v, 1 << 18, slots=2)
        st1 = sa.cache_stats()
        assert np.array_equal(c, want[0]) and np.array_equal(s, want[1])
        nchunks = -(-len(x) // (1 << 18))
        assert st1["misses"] - st0["misses"] == 2 * nchunks and st1["chunks"] == 2 * nchunks and st1["bytes"] == x.nbytes + v.nbytes
        c, s = _stream(x, v, 1 << 18, slots=2)
        st2 = sa.cache_stats()
        assert np.array_equal(c, want[0]) and np.array_equal(s, want[1])
        assert st2["hits"] - st1["hits"] == 2 * nchunks and st2["misses"] == st1["misses"]
        # a different chunking of the same columns: new pieces, same answer
        c, s = _stream(x, v, 300_000, slots=1)
        assert np.array_equal(c, want[0]) and np.array_equal(s, want[1])
        vaex_amd.uncache_columns()
        st3 = sa.cache_stats()
        assert st3["ranges"] == 0 and st3["chunks"] == 0 and st3["bytes"] == 0
        # unregistered again: changes of the array are seen
        x2 = x.copy()
        x[:] = 0.5
        c, s = _stream(x, v, 1 << 18)
        assert c.sum() == len(x) and c[36] == len(x)
        x[:] = x2


def test_cache_evicts_least_recently_used_chunks_within_its_budget():
    import vaex_amd
    x, v, sel = _data(1 << 20, 2)
    want = _expected(x, v, -4, 4, 64)
    sa.config_set("cache_bytes", 3 * (1 << 16) * 8)  # room for three 64 Ki-row chunks; the pass needs 32
    vaex_amd.cache_columns({"x": x, "v": v})
    for _ in range(2):
        c, s = _stream(x, v, 1 << 16, slots=2)
        assert np.array_equal(c, want[0]) and np.array_equal(s, want[1])
    st = sa.cache_stats()
    assert st["bytes"] <= 3 * (1 << 16) * 8 and st["evictions"] > 0
    sa.config_set("cache_bytes", 0)  # nothing fits: registered columns are still fed correctly
    sa.cache_clear()
    c, s = _stream(x, v, 1 << 16, slots=2)
    assert np.array_equal(c, want[0]) and np.array_equal(s, want[1])
    assert sa.cache_stats()["bytes"] == 0


def test_register_rejects_overlaps_and_unknown_ranges():
    x = np.zeros(1000)
    sa.cache_register(x, False)
    with pytest.raises(RuntimeError, match="overlaps"):
        sa.cache_register(x[10:20], False)
    sa.cache_unregister(x)
    with pytest.raises(RuntimeError, match="not a registered range"):
        sa.cache_unregister(x)


def test_frame_results_with_cached_columns_and_selection():
    import vaex_amd
    from vaex_amd.binned import Frame
    x, v, sel = _data(3_000_000, 3)
    y = np.random.default_rng(9).normal(0, 1, len(x))
    f = Frame(x=x, y=y, v=v, s=sel.astype(bool), nthreads=4, chunk_size=1 << 18)
    ref = f.mean("v", binby=["x", "y"], limits=[[-4, 4], [-4, 4]], shape=32, selection="s")
    vaex_amd.cache_columns({"x": x, "y": y, "v": v})
    for _ in range(2):
        got = f.mean("v", binby=["x", "y"], limits=[[-4, 4], [-4, 4]], shape=32, selection="s")
        assert np.allclose(got, ref, rtol=1e-12, atol=0, equal_nan=True)
    assert sa.cache_stats()["hits"] > 0


@pytest.mark.parametrize("mode", ["r", "c", "r+"])
def test_memory_mapped_files_stream_and_cache(tmp_path, mode):
    """north_star: "column chunks streamed from memory-mapped HDF5 / Arrow into pinned host buffers".  What vaex hands the aggregators
    for an opened file is a numpy view of a memory MAPPING (vaex/dataset.py:506-531 over np.memmap / mmap'ed HDF5 datasets): the columns
    here are real files mapped read-only ("r": PROT_READ pages, the way vaex.open maps them — page-locking them takes
    hipHostRegisterReadOnly), copy-on-write ("c") and read-write ("r+").  Streamed chunk by chunk, then registered with the device
    column cache (pin=True: the mapping itself is page-locked) and served from HBM on the second pass."""
    import vaex_amd
    x, v, sel = _data(1_500_001, 7)
    want = _expected(x, v, -4, 4, 64)
    fx, fv = tmp_path / "x.f8", tmp_path / "v.f8"
    x.tofile(fx); v.tofile(fv)
    mx = np.memmap(fx, dtype="f8", mode=mode, shape=(len(x),))
    mv = np.memmap(fv, dtype="f8", mode=mode, shape=(len(v),))
    assert isinstance(mx, np.memmap) and (mode != "r" or not mx.flags.writeable)
    for feeder in (1, 2):
        sa.config_set("feeder", feeder)
        c, s = _stream(mx, mv, 1 << 17, slots=2)
        assert np.array_equal(c, want[0]) and np.array_equal(s, want[1]), (mode, feeder)
    sa.config_set("feeder", 1)
    assert vaex_amd.cache_columns({"x": mx, "v": mv}, pin=True) == x.nbytes + v.nbytes
    st0 = sa.cache_stats()
    c, s = _stream(mx, mv, 1 << 18, slots=2)
    st1 = sa.cache_stats()
    assert np.array_equal(c, want[0]) and np.array_equal(s, want[1])
    assert st1["misses"] > st0["misses"]
    c, s = _stream(mx, mv, 1 << 18, slots=2)
    st2 = sa.cache_stats()
    assert np.array_equal(c, want[0]) and np.array_equal(s, want[1])
    assert st2["hits"] - st1["hits"] == 2 * -(-len(x) // (1 << 18)) and st2["misses"] == st1["misses"]
    vaex_amd.uncache_columns()
    del mx, mv


@pytest.mark.parametrize("threads", [0, 1, 3, 16])
def test_upload_moves_a_whole_pageable_column(threads):
    """round 4: vxh_upload — a whole host array into device memory the caller owns, pushed by several copy threads (what the wrapped
    df.groupby does with plain numpy columns, once per call).  Sizes that are no multiple of the slices or of the 32 MiB pieces, a
    one-byte dtype, a length-mismatch and a host destination"""
    import torch
    rng = np.random.default_rng(threads)
    for n, dt in ((70_000_001, np.int64), (33_554_433, np.uint8), (5, np.float64)):
        a = rng.integers(0, 250, n).astype(dt)
        t = torch.zeros(n, dtype={np.int64: torch.int64, np.uint8: torch.uint8, np.float64: torch.float64}[dt], device="cuda")
        sa.upload(a, t, threads)
        assert np.array_equal(t.cpu().numpy(), a)
    with pytest.raises(RuntimeError, match="byte length"):
        sa.upload(np.zeros(8, dtype=np.int64), torch.zeros(7, dtype=torch.int64, device="cuda"))
    with pytest.raises(RuntimeError, match="host array, device array"):
        sa.upload(np.zeros(8, dtype=np.int64), np.zeros(8, dtype=np.int64))
