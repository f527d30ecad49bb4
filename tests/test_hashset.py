"""vaex_amd.hashset.ordered_set_<dtype> against the reference's own compiled `superutils.ordered_set_<dtype>` (oracle/_ref),
method by method and key by key: update(return_values=True) incl. masked rows / NaN / -0.0 and +0.0, merge, create
(-> map_ordinal's dtype and ordinals), isin, null_index / nan_index, key_array / keys, flatten_values, pickling — plus the
cases of the reference's own tests/internal/hash_test.py:69-153 (test_set_bool, test_set_float over nan x missing x nmaps).
(hash_test.py:374-483 test `index_hash_<T>` — the row-index map behind df.join, vaex/join.py — which is not on the groupby path:
SURVEY §8 row a9 is `ordered_set<T>`; joins are §2 out of scope.)

Two backends run the same checks:
  * `-m gpu`: the product — the device hash map of libvaexhip.so behind the host logic of vaex_amd/hashset.py;
  * here (no GPU): the same host logic over a dict-backed stand-in for the device table (claim order = insertion order),
    so that the 270 lines of numpy in hashset.py are pinned against the reference on every CPU run.
Ordinals of a growing set are the product's own (claim order on the device): parity is per key — keys[ordinal of row] is the
row's key on both sides; sets made by `create` have their ordinals fixed by the key array and must agree exactly."""
import pickle

import numpy as np
import pytest

DTYPES = ["float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"]


class _DictMap:
    """stand-in for vaex_amd.superagg.ordered_set_int64 (tests without a GPU): int64 keys -> dense ordinals in insertion order"""

    def __init__(self, capacity_hint=0):
        self.d = {}

    def update(self, keys, mask=None):
        for k in np.asarray(keys, dtype=np.int64).tolist():
            self.d.setdefault(k, len(self.d))

    def set_keys(self, keys):
        self.d = {k: i for i, k in enumerate(np.asarray(keys, dtype=np.int64).tolist())}

    def map_ordinal(self, keys):
        return np.array([self.d.get(k, -1) for k in np.asarray(keys, dtype=np.int64).tolist()], dtype=np.int64)

    def key_array(self):
        return np.array(list(self.d), dtype=np.int64)

    def __len__(self):
        return len(self.d)


@pytest.fixture(params=["host-logic", pytest.param("hip", marks=pytest.mark.gpu)])
def hs(request, monkeypatch):
    from vaex_amd import hashset
    if request.param == "hip":
        import vaex_amd
        if vaex_amd.superagg.device_count() == 0:
            pytest.fail("no HIP device visible: -m gpu tests need the GPU box")
    else:
        class _Fake:
            ordered_set_int64 = _DictMap
        monkeypatch.setattr(hashset, "_sa", _Fake)
    return hashset


@pytest.fixture(scope="module")
def refu():
    from oracle import oracle
    m = oracle.ref_module("superutils")
    if m is None:
        pytest.skip("oracle/_ref/superutils not built")
    return m


def _keys(rng, name, n):
    if name == "bool":
        return rng.random(n) < 0.5
    if name.startswith("float"):
        a = (rng.integers(-40, 40, n) / 4).astype(name)
        a[rng.random(n) < 0.05] = np.nan
        a[rng.random(n) < 0.03] = -0.0
        a[rng.random(n) < 0.03] = 0.0
        return a
    info = np.iinfo(name)
    a = rng.integers(max(info.min, -60), min(info.max, 60), n).astype(name)
    a[:4] = [info.min, info.max, info.min, info.max]   # the extremes are ordinary keys (INT64_MIN is the device table's EMPTY marker)
    return a


def _same_key(a, b):
    """per-key equality the way the sets see keys: by bits for floats (NaN == NaN; -0.0 != +0.0, src/hash.hpp:138-150)"""
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return np.array_equal(a.view("u%d" % a.itemsize)[~np.isnan(a)], b.view("u%d" % b.itemsize)[~np.isnan(b)]) and np.array_equal(np.isnan(a), np.isnan(b))
    return np.array_equal(a, b)


def _keyset(s):
    """{bits of every real key} of a set (reference's or ours), null / NaN slots left out"""
    keys = np.asarray(s.key_array())
    live = np.ones(len(keys), bool)
    if s.has_null:
        live[s.null_index] = False
    if s.has_nan:
        live[s.nan_index] = False
    k = keys[live]
    return sorted(k.view("u%d" % k.itemsize).tolist()) if k.dtype.kind == "f" else sorted(k.tolist())


def _check_rows(s, ar, mask, ords):
    """ordinals returned for the rows of `ar` point at the rows' keys (null rows at null_index, NaN rows at nan_index)"""
    keys = np.asarray(s.key_array())
    ords = np.asarray(ords).astype(np.int64)
    assert ords.min() >= 0 and ords.max() < len(s)
    null = np.zeros(len(ar), bool) if mask is None else np.asarray(mask, bool)
    nan = (ar != ar) & ~null if ar.dtype.kind == "f" else np.zeros(len(ar), bool)
    if null.any():
        assert (ords[null] == s.null_index).all()
    if nan.any():
        assert (ords[nan] == s.nan_index).all()
    live = ~(null | nan)
    assert _same_key(keys[ords[live]], ar[live])


@pytest.mark.parametrize("name", DTYPES)
@pytest.mark.parametrize("masked", [False, True])
def test_update_return_values_per_key(hs, refu, name, masked):
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"hs-{name}-{masked}".encode()))
    n = 3000
    cs = 1024 * 1024
    ours, theirs = getattr(hs, "ordered_set_" + name)(1), getattr(refu, "ordered_set_" + name)(1)
    assert len(ours) == len(theirs) == 0 and ours.null_index == -1 and not ours.has_null and not ours.has_nan
    seen = []
    for part in range(3):   # several calls: the ordinals handed back by earlier calls must stay valid
        ar = _keys(rng, name, n)
        mask = (rng.random(n) < 0.1) if masked and part != 0 else None   # (the null key appears in the SECOND call: among the keys)
        if part == 0 and ar.dtype.kind == "f":
            ar[np.isnan(ar)] = 1.0                                          # (and NaN too)
        args = (ar,) + ((mask,) if mask is not None else ()) + (-1,)
        mine = ours.update(*args, chunk_size=cs, bucket_size=cs * 4, return_values=True)
        ref = theirs.update(*args, chunk_size=cs, bucket_size=cs * 4, return_values=True)
        assert isinstance(mine, tuple) and len(mine) == 2
        assert mine[0].dtype == ref[0].dtype and mine[1].dtype == ref[1].dtype and mine[0].shape == ref[0].shape and mine[1].shape == ref[1].shape
        out = np.empty(n, dtype="i8")
        flat = ours.flatten_values(mine[0], mine[1], out)
        _check_rows(theirs, ar, mask, theirs.flatten_values(ref[0], ref[1], np.empty(n, dtype="i8")))
        _check_rows(ours, ar, mask, flat)
        seen.append((ar, mask, np.array(flat)))
        assert len(ours) == len(theirs) and ours.count == theirs.count
        assert (ours.null_count, ours.nan_count, ours.has_null, ours.has_nan) == (theirs.null_count, theirs.nan_count, theirs.has_null, theirs.has_nan)
        assert (ours.null_index >= 0) == (theirs.has_null) and (ours.nan_index >= 0) == (theirs.has_nan)
        assert _keyset(ours) == _keyset(theirs)
    for ar, mask, flat in seen:    # ordinals of earlier calls still name the same keys
        _check_rows(ours, ar, mask, flat)
    # the update without return_values returns None on both sides
    assert ours.update(seen[0][0], -1, chunk_size=cs, bucket_size=cs * 4) is None and theirs.update(seen[0][0], -1, chunk_size=cs, bucket_size=cs * 4) is None
    # map_ordinal: narrowest integer type, known keys -> their ordinal, unknown -> -1, NaN -> nan_index (or -1)
    probe = np.concatenate([seen[0][0][:500], _keys(rng, name, 500)])
    a, b = ours.map_ordinal(probe), theirs.map_ordinal(probe)
    assert a.dtype == b.dtype and a.shape == b.shape
    assert np.array_equal(a >= 0, b >= 0)
    hit = a >= 0
    ka, kb = np.asarray(ours.key_array()), np.asarray(theirs.key_array())
    assert _same_key(ka[a[hit].astype("i8")], kb[b[hit].astype("i8")])
    assert np.array_equal(ours.isin(probe), theirs.isin(probe))
    if name.startswith("float"):   # a key the sets never saw, and the two zeros told apart
        unknown = np.array([1e30, -1e30], dtype=name)
        assert (ours.map_ordinal(unknown) == -1).all() and (theirs.map_ordinal(unknown) == -1).all() and not ours.isin(unknown).any()
    # keys(): python objects, None in the null slot
    k1, k2 = ours.keys(), theirs.keys()
    assert len(k1) == len(k2) and (None in k1) == (None in k2)
    if ours.has_null:
        assert k1[ours.null_index] is None and k2[theirs.null_index] is None
    # sealed sets refuse merges with the reference's message (src/hash_primitives.hpp:694-696)
    ours.seal(); theirs.seal()
    for mod, s in ((hs, ours), (refu, theirs)):
        with pytest.raises(RuntimeError, match="hashmap is sealed, cannot merge"):
            s.merge([getattr(mod, "ordered_set_" + name)(1)])


@pytest.mark.parametrize("name", DTYPES)
def test_create_fixes_the_ordinals_exactly(hs, refu, name):
    """ordered_set::create (src/hash_primitives.hpp:486-537): keys[i] gets ordinal i — vaex re-creates its sets from sorted
    key arrays (vaex/hash.py:260-283), so here map_ordinal / null_index / nan_index / dtype must EQUAL the reference's."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"create-{name}".encode()))
    ar = _keys(rng, name, 4000)
    uniq = np.unique(ar[~np.isnan(ar)] if ar.dtype.kind == "f" else ar)
    if ar.dtype.kind == "f":
        uniq = uniq[~((uniq == 0) & np.signbit(uniq))]    # (np.unique folds the zeros; keep +0.0)
    for with_null, with_nan in ((False, False), (True, False), (False, True), (True, True)):
        if with_nan and ar.dtype.kind != "f":
            continue
        keys = uniq.copy()
        nan_count = null_count = 0
        null_index = -1
        if with_nan:
            keys = np.concatenate([keys, np.array([np.nan], dtype=name)]); nan_count = 7
        if with_null:
            keys = np.concatenate([keys, np.zeros(1, dtype=name)]); null_index = len(keys) - 1; null_count = 3   # arrow sorts nulls last
        ours = getattr(hs, "ordered_set_" + name)(keys, null_index, nan_count, null_count, "fp")
        theirs = getattr(refu, "ordered_set_" + name)(keys, null_index, nan_count, null_count, "fp")
        assert len(ours) == len(theirs) == len(keys)
        assert (ours.null_index if with_null else -1) == (theirs.null_index if with_null else -1)
        assert (ours.nan_index if with_nan else -1) == (theirs.nan_index if with_nan else -1)
        assert (ours.null_count, ours.nan_count, ours.has_null, ours.has_nan, ours.fingerprint) == (theirs.null_count, theirs.nan_count, theirs.has_null, theirs.has_nan, theirs.fingerprint)
        probe = np.concatenate([ar, _keys(rng, name, 300)])
        a, b = ours.map_ordinal(probe), theirs.map_ordinal(probe)
        assert a.dtype == b.dtype and np.array_equal(a, b)
        assert np.array_equal(ours.isin(probe), theirs.isin(probe))
        assert _same_key(np.asarray(ours.key_array()), np.asarray(theirs.key_array())) or with_null   # (the null slot holds a placeholder)
        with pytest.raises(RuntimeError, match="sealed"):
            ours.update(ar, -1)
        # pickling goes through (type, (keys, null_index, nan_count, null_count, fingerprint)) — vaex/hash.py:21-25
        again = pickle.loads(pickle.dumps(ours))
        assert np.array_equal(again.map_ordinal(probe), b) and again.fingerprint == "fp" and len(again) == len(keys)
    if ar.dtype.kind == "f":   # the reference's consistency checks of create
        for s in (hs, refu):
            with pytest.raises(RuntimeError, match="NaN found in data, while claiming there should be none"):
                getattr(s, "ordered_set_" + name)(np.array([1, np.nan], dtype=name), -1, 0, 0, "")
            with pytest.raises(RuntimeError, match="no NaN found in data, while claiming there should be"):
                getattr(s, "ordered_set_" + name)(np.array([1, 2], dtype=name), -1, 1, 0, "")


@pytest.mark.parametrize("name", ["float64", "float32", "int64", "int8", "uint64", "bool"])
@pytest.mark.parametrize("specials", [False, True])
def test_merge_per_key(hs, refu, name, specials):
    """ordered_set::merge (src/hash_primitives.hpp:693-720) as TaskPartHashmapUniqueCreate.reduce calls it (vaex/cpu.py:366-385).
    With a null key / NaN in the merged-in sets the reference's merge only adds their COUNTS: it never assigns nan_value /
    null_value (key_array() then writes at index 0x7fffffff — a segfault here) and numbers new keys from maps.size() without
    the null / NaN offset — so for that case the reference is asked for its counts only and ours is checked per key."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(f"merge-{name}".encode()))
    cs = 1 << 20
    sets = []
    for mod in (hs, refu):
        rng2 = np.random.default_rng(zlib.crc32(f"merge-data-{name}".encode()))
        parts = []
        for i in range(3):
            s = getattr(mod, "ordered_set_" + name)(1)
            ar = _keys(rng2, name, 2000)
            if ar.dtype.kind == "f" and (i == 0 or not specials):
                ar[np.isnan(ar)] = 2.0
            mask = (rng2.random(2000) < 0.05) if (i == 2 and specials) else None
            s.update(*((ar,) + ((mask,) if mask is not None else ()) + (-1,)), chunk_size=cs, bucket_size=4 * cs)
            parts.append(s)
        parts[0].merge(parts[1:])
        sets.append(parts[0])
    ours, theirs = sets
    assert (ours.null_count, ours.nan_count) == (theirs.null_count, theirs.nan_count)
    assert (ours.has_null, ours.has_nan) == (theirs.has_null, theirs.has_nan)
    probe = _keys(rng, name, 1500)
    real = ~np.isnan(probe) if probe.dtype.kind == "f" else np.ones(len(probe), bool)
    a = ours.map_ordinal(probe)
    ka = np.asarray(ours.key_array())
    assert _same_key(ka[a[real & (a >= 0)].astype("i8")], probe[real & (a >= 0)])
    if not specials:
        assert _keyset(ours) == _keyset(theirs) and len(ours) == len(theirs)
        b = theirs.map_ordinal(probe)
        assert a.dtype == b.dtype and np.array_equal(a >= 0, b >= 0)
        assert np.array_equal(ours.isin(probe), theirs.isin(probe))
    else:
        assert ours.has_null and 0 <= ours.null_index < len(ours) and len(ours) == len(_keyset(ours)) + 1 + (1 if ours.has_nan else 0)
        if ours.has_nan:
            assert np.isnan(ka[ours.nan_index]) and (a[~real] == ours.nan_index).all() and ours.nan_index != ours.null_index
        # every key of the three parts is in the merged set
        rng3 = np.random.default_rng(zlib.crc32(f"merge-data-{name}".encode()))
        for i in range(2):
            ar = _keys(rng3, name, 2000)
            live = ~np.isnan(ar) if ar.dtype.kind == "f" else np.ones(len(ar), bool)
            assert (ours.map_ordinal(ar[live]) >= 0).all()


def test_concurrent_updates_of_one_set(hs):
    """vaex's pool threads call update() on ONE set at the same time (TaskPartHashmapUniqueCreate.process, vaex/cpu.py:340-361):
    the null key and NaN must get ONE ordinal each and every count must add up"""
    import threading
    s = hs.ordered_set_float64(1)
    rng = np.random.default_rng(8)
    parts = []
    for i in range(16):
        ar = rng.integers(0, 50, 5000).astype("f8")
        ar[::97] = np.nan
        parts.append((ar, rng.random(5000) < 0.02))
    start = threading.Barrier(8)

    def work(j):
        start.wait()
        for ar, mask in parts[j::8]:
            s.update(ar, mask, -1, chunk_size=1 << 20, bucket_size=1 << 22)
    threads = [threading.Thread(target=work, args=(j,)) for j in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert len(s) == 52 and s.has_null and s.has_nan and s.null_index != s.nan_index
    assert s.null_count == sum(int(m.sum()) for _, m in parts)
    assert s.nan_count == sum(int((np.isnan(a) & ~m).sum()) for a, m in parts)
    keys = s.key_array()
    assert np.isnan(keys[s.nan_index]) and sorted(np.delete(keys, [s.null_index, s.nan_index]).tolist()) == list(range(50))


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own tests (tests/internal/hash_test.py)
# ---------------------------------------------------------------------------------------------------------------------
def test_set_bool(hs):
    # hash_test.py:69-76
    bset = hs.ordered_set_bool(4)
    ar = np.array([True, True, False, False, True])
    chunk_size = 1024**2
    bset.update(ar, -1, chunk_size=chunk_size, bucket_size=chunk_size * 4)
    keys = bset.key_array()
    assert len(keys) == 2
    assert set(keys.tolist()) == {True, False}


def _dropnan(sequence, expect=None):
    # hash_test.py:14-24
    original_type = type(sequence)
    sequence = list(sequence)
    non_nan = [k for k in sequence if k == k]
    if expect is not None:
        assert len(sequence) - len(non_nan) == expect, "expected 1 nan value"
    return original_type(non_nan)


@pytest.mark.parametrize("nan", [False, True])
@pytest.mark.parametrize("missing", [False, True])
@pytest.mark.parametrize("nmaps", [1, 2, 3])
def test_set_float(hs, nan, missing, nmaps):
    # hash_test.py:79-153, statement by statement (repickle = pickle round trip)
    ar = np.arange(4, dtype='f8')[::-1].copy()
    keys_expected = [3, 2, 1, 0]
    null_index = 2
    if missing:
        mask = [0, 0, 1, 0]
        keys_expected[null_index] = None
    if nan:
        ar[1] = np.nan
        keys_expected[1] = np.nan
    oset = hs.ordered_set_float64(nmaps)
    if missing:
        ordinals_local, map_index = oset.update(ar, mask, return_values=True)
    else:
        ordinals_local, map_index = oset.update(ar, return_values=True)
    ordinals = np.empty(len(keys_expected), dtype='i8')
    ordinals = oset.flatten_values(ordinals_local, map_index, ordinals)
    keys = oset.keys()
    assert _dropnan(np.take(keys, ordinals).tolist()) == _dropnan(keys_expected)

    # plain object keys
    oset.seal()
    keys = oset.keys()
    expect_nan = 1 if nan else None
    assert _dropnan(set(keys), expect=expect_nan) == _dropnan(set(keys_expected), expect=expect_nan)
    # (the reference's line is `oset.map_ordinal(keys)` on the python list, None included — pybind11 converts it to a NaN)
    assert oset.map_ordinal(np.array([np.nan if k is None else k for k in keys], dtype='f8')).dtype.name == 'int8'

    # arrays
    keys = oset.key_array().tolist()
    if missing:
        keys[oset.null_index] = None
    assert _dropnan(set(keys), expect=expect_nan) == _dropnan(set(keys_expected), expect=expect_nan)
    if nan:
        assert np.isnan(keys[oset.nan_index])
    ordinals = oset.map_ordinal(np.array([np.nan if k is None else k for k in keys], dtype='f8')).tolist()
    if missing:
        ordinals[oset.null_index] = oset.null_index
    assert ordinals == list(range(4))

    # tests extraction and constructor
    keys = oset.key_array()
    set_copy = hs.ordered_set_float64(keys, oset.null_index, oset.nan_count, oset.null_count, '')
    keys = set_copy.key_array().tolist()
    if missing:
        keys[oset.null_index] = None
    assert _dropnan(set(keys)) == _dropnan(set(keys_expected))
    if nan:
        assert np.isnan(keys[set_copy.nan_index])
    ordinals = set_copy.map_ordinal(np.array([np.nan if k is None else k for k in keys], dtype='f8')).tolist()
    if missing:
        ordinals[set_copy.null_index] = set_copy.null_index
    assert ordinals == list(range(4))

    # test pickle
    set_copy = pickle.loads(pickle.dumps(oset))
    keys = set_copy.key_array().tolist()
    if missing:
        keys[oset.null_index] = None
    assert _dropnan(set(keys)) == _dropnan(set(keys_expected))
    if nan:
        assert np.isnan(keys[set_copy.nan_index])
    ordinals = set_copy.map_ordinal(np.array([np.nan if k is None else k for k in keys], dtype='f8')).tolist()
    if missing:
        ordinals[set_copy.null_index] = set_copy.null_index
    assert ordinals == list(range(4))
