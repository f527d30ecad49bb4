"""Lifetime of HBM-resident columns (include/vaex_hip.h "Data pointers", VXH_MEM_DEVICE): Grid.bin over device columns returns with its
kernels enqueued on the slot's non-blocking stream, so the columns are read AFTER the call has returned.  The reference's bin() is
synchronous (src/agg.hpp:84-137) and its callers keep their arrays alive for the call only (vaex/cpu.py:708-710) — a Python caller that
does the same with torch tensors hands the blocks back to torch's caching allocator while they are being read (VERDICT r5 weak #1c).
The shim therefore keeps a reference to every device array until the slot that read it is idle; a column produced on a non-blocking
side stream is ordered with wait_stream."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sa = pytest.importorskip("vaex_amd.superagg")


def _numpy_count(x, y, shape):
    """the scalar binner's cells (src/binners.cpp:13-57): [nan, underflow, bins..., overflow] per dimension, dim 0 fastest"""
    def sub(v):
        s = (v - (-4.0)) * (1.0 / 8.0)
        i = np.where(np.isnan(v), 0, np.where(s < 0, 1, np.where(s >= 1, shape + 2, (s * shape).astype(np.int64) + 2)))
        return i
    flat = sub(x) + sub(y) * (shape + 3)
    return np.bincount(flat, minlength=(shape + 3) ** 2).reshape(shape + 3, shape + 3).T


@pytest.mark.parametrize("rounds", [6])
def test_columns_freed_and_overwritten_between_bin_and_get_result(rounds):
    """free + reallocate + overwrite between bin() and get_result(): the grid must be that of the columns as they were handed over"""
    import torch
    n, shape = 40_000_000, 256
    g = torch.Generator(device="cuda").manual_seed(11)
    for r in range(rounds):
        x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        xh, yh = x.cpu().numpy(), y.cpu().numpy()
        torch.cuda.synchronize()
        bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, shape)
        by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, shape)
        grid = sa.Grid([bx, by])
        count = sa.AggCount_int64(grid, 1, 1)
        bx.set_data(0, x); by.set_data(0, y)
        bx.clear_data_mask(0); by.clear_data_mask(0); count.clear_data_mask(0)
        ptrs = (x.data_ptr(), y.data_ptr())
        grid.bin(0, [count], n)
        # the caller lets go, as the reference's contract allows, and at once asks torch for blocks of the same size and fills them
        del x, y
        if r % 2:
            bx.set_data(0, torch.zeros(8, dtype=torch.float64, device="cuda"))   # (replaced: the old reference is RETIRED, not dropped)
            by.set_data(0, torch.zeros(8, dtype=torch.float64, device="cuda"))
        gc.collect()
        junk = [torch.full((n,), 100.0 + i, dtype=torch.float64, device="cuda") for i in range(2)]
        reused = {t.data_ptr() for t in junk} & set(ptrs)
        got = np.asarray(count.get_result())
        assert not reused or not sa.slot_busy(0), "torch handed out a block the slot may still read"
        want = _numpy_count(xh, yh, shape)
        assert int(got.sum()) == n and np.array_equal(got, want), (r, int(np.abs(got - want).sum()), bool(reused))
        del junk
    sa.synchronize()
    assert sa.retired_device_arrays(0) == 0


def test_retired_references_are_dropped_when_the_slot_is_idle():
    import torch
    import weakref
    n = 8_000_000
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 64)
    grid = sa.Grid([bx])
    count = sa.AggCount_int64(grid, 1, 1)
    bx.set_data(0, x); bx.clear_data_mask(0); count.clear_data_mask(0)
    alive = weakref.ref(x)
    del x
    gc.collect()
    assert alive() is not None                  # held by the binner
    grid.bin(0, [count], n)
    bx.set_data(0, torch.zeros(4, dtype=torch.float64, device="cuda"))
    sa.slot_wait(0)                             # drained: everything retired for slot 0 goes
    gc.collect()
    assert alive() is None and sa.retired_device_arrays(0) == 0 and not sa.slot_busy(0)
    # many chunks over one slot: the retired list stays bounded
    for i in range(400):
        t = torch.randn(100_000, dtype=torch.float64, device="cuda")
        bx.set_data(0, t)
        grid.bin(0, [count], len(t))
        del t
        assert sa.retired_device_arrays(0) <= 129
    assert int(np.asarray(count.get_result()).sum()) == n + 400 * 100_000


def test_a_column_written_on_a_side_stream_is_ordered_by_wait_stream():
    """the slot is ordered after the legacy default stream only; a non-blocking producer stream is the caller's to order (wait_stream)"""
    import torch
    n, shape = 60_000_000, 64
    side = torch.cuda.Stream()                 # (torch's pool streams are non-blocking)
    for r in range(4):
        x = torch.empty(n, dtype=torch.float64, device="cuda")
        x.fill_(-100.0)                          # on the default stream: every row in the underflow cell
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(3):                   # a producer that takes a while
                x.normal_()
            x.clamp_(-3.9, 3.9)                  # every row inside the limits
        bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, shape)
        grid = sa.Grid([bx])
        count = sa.AggCount_int64(grid, 1, 1)
        bx.set_data(0, x); bx.clear_data_mask(0); count.clear_data_mask(0)
        sa.wait_stream(0, side.cuda_stream)
        grid.bin(0, [count], n)
        got = np.asarray(count.get_result())
        assert int(got[2:-1].sum()) == n and int(got[1]) == 0, (r, got[:3], int(got.sum()))
        side.synchronize()
