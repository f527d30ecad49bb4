"""BinnerHash_<T>[_non_native](threads, expression, hashmap) with the reference's constructor and cells (SURVEY §8 row a10,
src/binner_hash.cpp:13-71): our class on a vaex_amd.hashset.ordered_set_<T> against the reference's own compiled
`superagg.BinnerHash_<T>` on the reference's `superutils.ordered_set_<T>` (oracle/_ref), both filled by the same calls.

  * sets made by `create` (ordinals = positions in the key array on both sides, the null key in the middle, NaN among the keys):
    the grids must be equal cell by cell;
  * growing sets (`update` with masked rows and NaNs): ordinals are each side's own, so the comparison is per KEY — the row count /
    the sum of the cell whose ordinal belongs to a key, the null key and NaN included;
  * what the reference cannot be asked (it reads map_many's -1 through its unsigned index type and writes one cell past the grid,
    src/binner_hash.cpp:36-40 — heap corruption): keys the set does not hold, NaN next to a set without one, masked rows next to a
    set without a null -> cell 0 ("invalid" in the reference's own comment, :10), checked against numpy;
  * shape / hash_bins / copy / pickle / byte-swapped columns / a second (scalar) dimension / device-resident columns.
The library's other layout ([unknown, bins, null] on a bare device table, used by vaex_amd.binned) keeps its tests in
test_gpu_parity.py / test_gpu_baseline_shapes.py."""
import pickle

import numpy as np
import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu

DTYPES = ["float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"]


@pytest.fixture(scope="module")
def sa():
    import vaex_amd
    if vaex_amd.superagg.device_count() == 0:
        pytest.fail("no HIP device visible: -m gpu tests need the GPU box")
    return vaex_amd.superagg


@pytest.fixture(scope="module")
def ref():
    ru, ra = oracle.ref_module("superutils"), oracle.ref_module("superagg")
    if ru is None or ra is None:
        pytest.skip("oracle/_ref not built")
    return ru, ra


def _keys(rng, name, n, distinct=200):
    if name == "bool":
        return rng.random(n) < 0.5
    dt = np.dtype(name)
    if dt.kind == "f":
        pool = np.concatenate([rng.normal(0, 100, distinct - 4), [0.0, -0.0, np.inf, -np.inf]]).astype(dt)
        return pool[rng.integers(0, len(pool), n)]
    info = np.iinfo(dt)
    pool = np.unique(np.concatenate([rng.integers(info.min, info.max, distinct - 2, dtype=dt, endpoint=True), np.array([info.min, info.max], dtype=dt)]))
    return pool[rng.integers(0, len(pool), n)]


def _run(mod, binner_cls, hashmap, data, mask, value, threads=1, extra=None):
    """count(*) and sum(value) over BinnerHash(hashmap) [x extra binner]"""
    b = binner_cls(threads, "k", hashmap)
    b.set_data(0, data)
    if mask is not None:
        b.set_data_mask(0, mask)
    binners = [b] + ([extra] if extra is not None else [])
    grid = mod.Grid(binners)
    cnt = mod.AggCount_float64(grid, 1, 1)
    tot = mod.AggSum_float64(grid, 1, 1)
    tot.set_data(0, value, 0)
    grid.bin(0, [cnt, tot], len(data))
    return np.array(cnt, copy=True).reshape(-1), np.array(tot, copy=True).reshape(-1), b


def _per_key(counts, sums, s):
    """cell contents keyed by what the cell's ordinal stands for in set `s`"""
    keys = np.asarray(s.key_array())
    out = {}
    for cell in range(1, len(keys) + 1):
        o = cell - 1
        if s.has_null and o == s.null_index:
            k = "null"
        elif s.has_nan and o == s.nan_index:
            k = "nan"
        else:
            k = keys[o].tobytes()  # (bit pattern: -0.0 and 0.0 are two keys on both sides)
        out[k] = (int(counts[cell]), float(sums[cell]))
    return out


@pytest.mark.parametrize("name", DTYPES)
def test_growing_set_per_key(sa, ref, name):
    from vaex_amd import hashset
    ru, ra = ref
    rng = np.random.default_rng(DTYPES.index(name))
    n = 20_000
    data = _keys(rng, name, n)
    if np.dtype(name).kind == "f":
        data[rng.random(n) < 0.02] = np.nan
    mask = rng.random(n) < 0.03
    value = rng.normal(3, 2, n)
    ours, theirs = getattr(hashset, "ordered_set_" + name)(1), getattr(ru, "ordered_set_" + name)(1)
    for lo in range(0, n, 7000):  # the null key and NaN are first seen in different calls: their ordinals sit among the keys'
        hi = min(n, lo + 7000)
        m = mask[lo:hi] if lo else np.zeros(hi - lo, dtype=bool)
        d = data[lo:hi].copy()
        if lo == 0 and d.dtype.kind == "f":
            d[d != d] = 1.0
        ours.update(d, m)
        theirs.update(d, m)
    assert len(ours) == len(theirs) and ours.null_index >= 0 and theirs.null_index >= 0
    # rows: every key is known to the sets (the reference must not see an unknown one), masked rows, NaNs
    data2 = data.copy()
    if data2.dtype.kind == "f":
        first = data[:7000]
        data2[:7000] = np.where(first != first, 1.0, first)
    m8 = mask.astype(np.uint8); m8[:7000] = 0
    c1, s1, b1 = _run(sa, getattr(sa, "BinnerHash_" + name), ours, data2, m8, value)
    c2, s2, b2 = _run(ra, getattr(ra, "BinnerHash_" + name), theirs, data2, m8, value)
    assert len(b1) == len(b2) == len(theirs) + 2 and b1.hash_bins == b2.hash_bins == len(theirs)
    assert c1[0] == c2[0] == 0 and c1[-1] == c2[-1] == 0 and c1.sum() == n
    got, want = _per_key(c1, s1, ours), _per_key(c2, s2, theirs)
    assert got.keys() == want.keys()
    for k in want:
        assert got[k][0] == want[k][0], k
        assert abs(got[k][1] - want[k][1]) <= 1e-12 * 12.0 * max(want[k][0], 1), k


@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("name", DTYPES)
def test_created_set_cell_by_cell(sa, ref, name, flip):
    """ordered_set_T(keys, null_index, nan_count, null_count, fingerprint) — vaex seals and re-creates its sets from sorted key arrays
    (vaex/hash.py:260-283): the ordinals are the positions, so the grids agree cell by cell"""
    from vaex_amd import hashset
    ru, ra = ref
    if flip and np.dtype(name).itemsize == 1:
        pytest.skip("one-byte dtypes have no byte order")
    rng = np.random.default_rng(100 + DTYPES.index(name))
    keys = np.unique(_keys(rng, name, 3000))
    isf = keys.dtype.kind == "f"
    if isf:
        keys = keys[keys == keys]
        keys = np.concatenate([keys[:5], [np.nan], keys[5:]]).astype(keys.dtype)
    null_at = 1 if len(keys) > 2 else -1
    if null_at >= 0:
        keys = np.concatenate([keys[:null_at], keys[:1], keys[null_at:]])  # a placeholder where the null key sits
        if name == "bool":
            keys = np.array([False, False, True])
    args = (keys, null_at, 1 if isf else 0, 7 if null_at >= 0 else 0, "fp")
    ours, theirs = getattr(hashset, "ordered_set_" + name)(*args), getattr(ru, "ordered_set_" + name)(*args)
    # (the reference's byte-swapping to_bins indexes its block-local `flipped` copy with the call's row numbers, src/binner_hash.cpp:28-38:
    #  beyond the first block of INDEX_BLOCK_SIZE = 1024 rows it reads past it — it can only be asked about one block)
    n = 1000 if flip else 30_000
    live = np.ones(len(keys), dtype=bool)
    if null_at >= 0:
        live[null_at] = False
    data = keys[live][rng.integers(0, live.sum(), n)]
    mask = (rng.random(n) < 0.05).astype(np.uint8) if null_at >= 0 else None
    value = rng.normal(-1, 3, n)
    col = data.astype(data.dtype.newbyteorder()) if flip else data
    post = name + ("_non_native" if flip else "")
    c1, s1, b1 = _run(sa, getattr(sa, "BinnerHash_" + post), ours, col, mask, value)
    c2, s2, b2 = _run(ra, getattr(ra, "BinnerHash_" + post), theirs, col, mask, value)
    assert len(b1) == len(b2) == len(keys) + 2
    np.testing.assert_array_equal(c1, c2)
    assert c1.sum() == n and c1[0] == 0
    if null_at >= 0:
        assert c1[null_at + 1] == int(mask.sum())
    if flip:  # ... and more rows than one block against the native byte order
        data = keys[live][rng.integers(0, live.sum(), 50_000)]
        mask = (rng.random(50_000) < 0.05).astype(np.uint8) if null_at >= 0 else None
        value = rng.normal(-1, 3, 50_000)
        c1, s1, _ = _run(sa, getattr(sa, "BinnerHash_" + post), ours, data.astype(data.dtype.newbyteorder()), mask, value)
        c2, s2, _ = _run(sa, getattr(sa, "BinnerHash_" + name), ours, data, mask, value)
        np.testing.assert_array_equal(c1, c2)
        assert np.all(np.abs(s1 - s2) <= 1e-12 * 16.0 * np.maximum(c2, 1))  # (float64 adds in whatever order the device ran them)
        assert c1.sum() == 50_000
    elif isf:
        assert c1[6 + (1 if null_at >= 0 else 0)] == int(((data != data) & ((mask == 0) if mask is not None else True)).sum()) > 0
    assert np.all(np.abs(s1 - s2) <= 1e-12 * 16.0 * np.maximum(c2, 1))


@pytest.mark.parametrize("name", ["int64", "float64", "float32", "uint8"])
def test_rows_the_reference_cannot_take_go_to_cell_zero(sa, name):
    """unknown keys, NaN next to a set that never saw one, masked rows next to a set without a null: the reference's to_bins reads
    map_many's -1 (src/hash_primitives.hpp:577,:584) as an unsigned index and adds at hash_bins + 2, one cell past the grid
    (src/binner_hash.cpp:36-40,:56-60); here they land in cell 0, the cell its comment calls `invalid` (:10)"""
    from vaex_amd import hashset
    rng = np.random.default_rng(5)
    dt = np.dtype(name)
    known = np.array([3, 5, 9, 17, 100], dtype=dt)
    s = getattr(hashset, "ordered_set_" + name)(1)
    s.update(known)
    assert s.null_index == -1 and len(s) == 5
    n = 10_000
    data = np.array([3, 5, 9, 17, 100, 4, 6, 200], dtype=dt)[rng.integers(0, 8, n)]
    if dt.kind == "f":
        data[::50] = np.nan
    mask = (rng.random(n) < 0.1).astype(np.uint8)
    value = np.ones(n)
    c, t, b = _run(sa, getattr(sa, "BinnerHash_" + name), s, data, mask, value)
    assert len(b) == 7 and b.hash_bins == 5
    ords = np.asarray(s.map_ordinal(data)).astype(np.int64)
    cell = np.where((mask == 1) | (data != data) | (ords < 0), 0, ords + 1)
    np.testing.assert_array_equal(c, np.bincount(cell, minlength=7))
    np.testing.assert_array_equal(t, np.bincount(cell, minlength=7).astype(float))
    assert c[0] > 0 and c[6] == 0


def test_copy_pickle_second_dimension_and_device_columns(sa, ref):
    import torch
    from vaex_amd import hashset
    ru, ra = ref
    rng = np.random.default_rng(9)
    keys = np.array([40, -7, 0, 12, 99, 5], dtype="i8")
    args = (keys, 2, 0, 3, "fp")  # the null key at position 2
    ours, theirs = hashset.ordered_set_int64(*args), ru.ordered_set_int64(*args)
    n = 50_000
    data = keys[np.array([0, 1, 3, 4, 5])][rng.integers(0, 5, n)]
    mask = (rng.random(n) < 0.1).astype(np.uint8)
    x = rng.normal(0, 1, n)
    value = rng.normal(0, 1, n)
    c1, s1, b1 = _run(sa, sa.BinnerHash_int64, ours, data, mask, value, extra=_scalar(sa, x))
    c2, s2, b2 = _run(ra, ra.BinnerHash_int64, theirs, data, mask, value, extra=_scalar(ra, x))
    assert len(c2) == 8 * 19
    np.testing.assert_array_equal(c1, c2)
    assert np.all(np.abs(s1 - s2) <= 1e-12 * 8.0 * np.maximum(c2, 1))
    position = {int(k): i for i, k in enumerate(keys)}
    cell = np.where(mask == 1, 2 + 1, np.array([position[int(k)] for k in data]) + 1)
    want = np.bincount(cell, minlength=8)
    assert want[0] == 0 and want[7] == 0 and want[3] == int(mask.sum())

    def count(binner, d, m):
        binner.set_data(0, d); binner.set_data_mask(0, m)
        grid = sa.Grid([binner]); a = sa.AggCount_int64(grid, 1, 1); grid.bin(0, [a], n)
        return np.array(a).reshape(-1)
    # copy() and pickle keep the set (src/binner_hash.cpp:21, :159-170)
    for clone in (b1.copy(), pickle.loads(pickle.dumps(b1))):
        assert type(clone) is type(b1) and len(clone) == len(b1) == 8 and clone.expression == "k" and clone.hash_bins == 6
        np.testing.assert_array_equal(count(clone, data, mask), want)
    # device-resident columns
    np.testing.assert_array_equal(count(sa.BinnerHash_int64(1, "k", ours), torch.from_numpy(data).cuda(), torch.from_numpy(mask).cuda()), want)


def _scalar(mod, x):
    b = mod.BinnerScalar_float64(1, "x", -3.0, 3.0, 16)
    b.set_data(0, x)
    return b


def test_large_set_with_null_between_the_keys(sa):
    """1e6 keys, the null key first seen after half of them (its ordinal sits in the middle: every later ordinal is shifted)"""
    from vaex_amd import hashset
    rng = np.random.default_rng(3)
    keys = rng.permutation(4_000_000)[:1_000_000].astype("i8") * 977 - 10**9
    s = hashset.ordered_set_int64(1)
    s.update(keys[:500_000])
    s.update(keys[500_000:500_010], np.array([1] + [0] * 9, dtype=bool))
    s.update(keys[500_000:])
    assert len(s) == 1_000_001 and s.null_index == 500_009
    n = 5_000_000
    rows = rng.integers(0, len(keys), n)
    data = keys[rows]
    mask = (rng.random(n) < 0.01).astype(np.uint8)
    b = sa.BinnerHash_int64(1, "k", s)
    b.set_data(0, data); b.set_data_mask(0, mask)
    grid = sa.Grid([b]); a = sa.AggCount_int64(grid, 1, 1); grid.bin(0, [a], n)
    got = np.array(a).reshape(-1)
    ords = np.asarray(s.map_ordinal(data)).astype(np.int64)
    assert ords.min() >= 0
    want = np.bincount(np.where(mask == 1, s.null_index + 1, ords + 1), minlength=len(s) + 2)
    np.testing.assert_array_equal(got, want)
    ka = np.asarray(s.key_array())
    np.testing.assert_array_equal(ka[ords[mask == 0][:1000]], data[mask == 0][:1000])
