"""`ordered_set_<dtype>` of vaex.superutils on the GPU hash map — the class surface vaex's groupby and hashing code
calls (/root/reference/packages/vaex-core/src/hash_primitives.cpp:3-63 bindings; callers: vaex/hash.py:58-260
HashMapUnique, vaex/cpu.py:285-404 TaskPartHashmapUniqueCreate):

    ordered_set_T(nmaps, limit=-1)                                    src/hash_primitives.hpp:436-470
    ordered_set_T(keys, null_index, nan_count, null_count, fingerprint)   ::create, :486-537 (keys[i] gets ordinal i)
    update(values[, masks], start_index=0, chunk_size=, bucket_size=, return_values=False)   :98-295
    merge(others)   key_array()   keys()   __len__   map_ordinal(values)   isin(values)   flatten_values(...)
    null_index  nan_index  null_count  nan_count  has_null  has_nan  fingerprint  seal()  __sizeof__

Keys live in ONE open-addressing table in HBM (vaex_amd/csrc/vxh_hashmap.hip, the `ordered_set` of the pybind shim);
this module adds, in numpy on the host, what that table does not know about: float keys (a float64 / float32 key goes in
as its bit pattern — -0.0 and +0.0 are two keys, as for the reference's hash of the bits, src/hash.hpp:138-150;
NaN has its own ordinal, like the reference's nan_value), the ordinal of
the null key, ordinals fixed by the caller (the `create` constructor: vaex seals and re-creates its sets from sorted key
arrays, vaex/hash.py:260-283), and map_ordinal's narrowest-integer result type (:611-626).

Ordinals of a set that is still being filled are dense in claim order on the device (the reference: shard offset +
insertion order); the null key and NaN take the next free ordinal when they are first seen and keep it, like the
reference's add_null / add_nan (:455-470) — so the ordinals update(return_values=True) hands back stay valid while the set
grows.  Parity with the reference is per key, never per ordinal (except for sets made by `create`, whose ordinals are the
positions in the key array on both sides).
"""
import threading

import numpy as np

from . import superagg as _sa

_INT = ("int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool")
_FLOAT = ("float64", "float32")


def _narrowest(size):
    # src/hash_primitives.hpp:611-626
    if size < (1 << 7):
        return np.int8
    if size < (1 << 15):
        return np.int16
    if size < (1 << 31):
        return np.int32
    return np.int64


class _OrderedSet:
    """base of the generated ordered_set_<dtype> classes"""

    dtype_name = "int64"

    def __init__(self, *args, **kwargs):
        self.fingerprint = ""
        self.null_count = 0
        self.nan_count = 0
        self._specials = []     # [(keys on the device when first seen, "null" | "nan")]: ordinal = that count + position in this list
        self._fixed = None      # (key array incl. placeholders, null_index, nan_index) of a set made by `create`
        self._perm = None       # create: position in the key array of the device map's ordinal
        self.sealed = False
        self._np = np.dtype(self.dtype_name)
        self._is_float = self.dtype_name in _FLOAT
        self._limit = -1
        self._lock = threading.Lock()   # vaex's pool threads update ONE set concurrently (vaex/cpu.py:340-361; the reference: a mutex per map)
        if args and not isinstance(args[0], (int, np.integer)):
            self._create(*args, **kwargs)
        else:
            nmaps = args[0] if args else kwargs.get("nmaps", 1)
            self._limit = args[1] if len(args) > 1 else kwargs.get("limit", -1)
            del nmaps  # one table on the device: nothing to shard
            self._map = _sa.ordered_set_int64()

    # ---------------------------------------------------------------- key <-> table representation
    def _bits(self, ar):
        """keys as the int64 the device table stores, plus the mask of NaNs (floats)"""
        ar = np.asarray(ar)
        if self._is_float:
            a = np.ascontiguousarray(ar, dtype=self._np)
            nan = a != a
            bits = a.view(np.int64 if self._np.itemsize == 8 else np.int32).astype(np.int64)
            return bits, (nan if nan.any() else None)
        return np.ascontiguousarray(ar).astype(np.int64, copy=False) if ar.dtype != np.uint64 else np.ascontiguousarray(ar).view(np.int64), None

    def _unbits(self, bits):
        if self._is_float:
            return bits.astype(np.int64 if self._np.itemsize == 8 else np.int32).view(self._np)
        if self._np == np.uint64:
            return bits.view(np.uint64)
        return bits.astype(self._np)

    # ---------------------------------------------------------------- construction from a key array
    def _create(self, keys, null_index=-1, nan_count=0, null_count=0, fingerprint=""):
        if type(keys).__module__.startswith("pyarrow"):  # vaex/hash.py:260-283 hands sorted arrow arrays (null last) over
            if keys.null_count:
                keys = keys.fill_null(0)
            keys = keys.to_numpy(zero_copy_only=False)
        keys = np.ma.getdata(keys) if np.ma.isMaskedArray(keys) else np.asarray(keys)
        keys = np.ascontiguousarray(keys.astype(self._np, copy=False))
        self.fingerprint = fingerprint
        self.null_count = int(null_count)
        self.nan_count = int(nan_count)
        null_index = int(np.asarray(null_index).ravel()[0]) if np.ndim(null_index) else int(null_index)
        null_seen = self.null_count > 0 and null_index >= 0
        n = len(keys)
        real = np.ones(n, dtype=bool)
        nan_index = -1
        if self._is_float and self.nan_count > 0:
            where = np.nonzero(keys != keys)[0]
            if len(where):
                nan_index = int(where[0])
                real[where] = False
        if self._is_float and (self.nan_count > 0) != bool((keys != keys).any()):  # src/hash_primitives.hpp:505-513
            raise RuntimeError("no NaN found in data, while claiming there should be" if self.nan_count > 0 else "NaN found in data, while claiming there should be none")
        if null_seen:
            real[null_index] = False
        pos = np.nonzero(real)[0]
        bits, _ = self._bits(keys[pos])
        self._map = _sa.ordered_set_int64(len(bits))
        if len(bits):
            self._map.set_keys(bits)  # ordinal i <-> bits[i] <-> position pos[i] of the key array
        self._perm = pos if len(pos) != n else None
        self._fixed = (keys, null_index if null_seen else -1, nan_index)
        self.sealed = True

    # ---------------------------------------------------------------- filling
    def update(self, values, *args, **kwargs):
        """update(values[, masks], start_index, chunk_size, bucket_size, return_values) — src/hash_primitives.hpp:98-295.
        return_values=True hands back (ordinal of every row, map index of every row = 0: one table): masked rows get the null
        key's ordinal, NaNs the NaN's."""
        if self._fixed is not None:  # (the reference checks `sealed` in merge only, :694-696; a set made by `create` holds its
            raise RuntimeError("hashmap is sealed, cannot update")  # keys' positions and cannot take new ones here)
        masks = None
        rest = list(args)
        if rest and isinstance(rest[0], (np.ndarray, list, tuple)):
            masks = rest.pop(0)
        masks = kwargs.get("masks", masks)
        return_values = bool(kwargs.get("return_values", rest[3] if len(rest) > 3 else False))
        if return_values and self._limit is not None and self._limit >= 0:
            raise RuntimeError("Cannot combine limit with return_inverse")
        data = np.ma.getdata(values) if np.ma.isMaskedArray(values) else np.asarray(values)
        if np.ma.isMaskedArray(values) and masks is None:
            masks = np.ma.getmaskarray(values)
        bits, nan = self._bits(data)
        null = None
        if masks is not None:
            null = np.ascontiguousarray(masks).astype(bool, copy=False)
            if not null.any():
                null = None
        if nan is not None and null is not None:
            nan = nan & ~null
            if not nan.any():
                nan = None
        special = null if nan is None else (nan if null is None else (null | nan))
        live_bits = bits if special is None else np.ascontiguousarray(bits[~special])
        with self._lock:
            if len(live_bits):
                self._map.update(live_bits)
            # the null key and NaN take the next free ordinal at first sight, nulls before NaNs (:262-277), behind the call's keys
            if null is not None:
                if not self.null_count:
                    self._specials.append((self._n_keys(), "null"))
                self.null_count += int(null.sum())
            if nan is not None:
                if not self.nan_count:
                    self._specials.append((self._n_keys(), "nan"))
                self.nan_count += int(nan.sum())
        if not return_values:
            return None
        out = np.empty(len(bits), dtype=np.int64)
        if special is None:
            out[...] = self._public(np.asarray(self._map.map_ordinal(live_bits))) if len(live_bits) else 0
        else:
            if len(live_bits):
                out[~special] = self._public(np.asarray(self._map.map_ordinal(live_bits)))
            if null is not None:
                out[null] = self.null_index
            if nan is not None:
                out[nan] = self.nan_index
        return out, np.zeros(len(bits), dtype=np.int16)

    def merge(self, others):
        if self.sealed:
            raise RuntimeError("hashmap is sealed, cannot merge")
        for other in others:
            other = getattr(other, "_internal", other)  # a HashMapUnique wrapper or the set itself
            if other is self:
                continue
            keys = np.asarray(other.key_array())
            live = np.ones(len(keys), dtype=bool)
            if other.has_null:
                live[other.null_index] = False
            if other.has_nan:
                live[other.nan_index] = False
            bits, _ = self._bits(keys[live])
            if len(bits):
                self._map.update(bits)
            if other.null_count:
                if not self.null_count:
                    self._specials.append((self._n_keys(), "null"))
                self.null_count += other.null_count
            if other.nan_count:
                if not self.nan_count:
                    self._specials.append((self._n_keys(), "nan"))
                self.nan_count += other.nan_count

    def seal(self):
        self.sealed = True

    # ---------------------------------------------------------------- reading
    def _n_keys(self):
        return len(self._map)

    def _public(self, ords):
        """device ordinals (dense over the real keys) -> the set's ordinals (the null key and NaN sit among them)"""
        ords = np.asarray(ords, dtype=np.int64)
        if not self._specials:
            return ords
        out = ords.copy()
        for t, _ in self._specials:
            out += (ords >= t)
        return np.where(ords >= 0, out, -1)

    def _special_index(self, kind):
        for i, (t, k) in enumerate(self._specials):
            if k == kind:
                return t + i
        return -1

    def __len__(self):
        if self._fixed is not None:
            return len(self._fixed[0])
        return self._n_keys() + len(self._specials)

    @property
    def count(self):
        return len(self)

    @property
    def has_null(self):
        return self.null_index >= 0 if self._fixed is not None else self.null_count > 0

    @property
    def has_nan(self):
        return self.nan_count > 0

    @property
    def null_index(self):
        if self._fixed is not None:
            return self._fixed[1]
        return self._special_index("null")

    @property
    def nan_index(self):
        if self._fixed is not None:
            return self._fixed[2]
        return self._special_index("nan")

    def _binner_view(self):
        """what BinnerHash_<T>(threads, expression, this set) needs (src/binner_hash.cpp:13-20 reads hashmap->size(), null_index(),
        nan_index()): (device table, the set's ordinal of every device ordinal — None when they are the same —, len(self),
        null_index, nan_index when a NaN was seen else -1)"""
        with self._lock:
            if self._perm is not None:
                public = np.ascontiguousarray(self._perm, dtype=np.int64)
            elif self._fixed is None and self._specials:
                public = np.ascontiguousarray(self._public(np.arange(self._n_keys(), dtype=np.int64)))
            else:
                public = None
            return self._map, public, len(self), int(self.null_index), int(self.nan_index) if self.nan_count > 0 else -1

    def key_array(self):
        """keys ordered by ordinal; the null key's slot holds a placeholder, NaN's slot NaN (src/hash_primitives.hpp:303-328)"""
        if self._fixed is not None:
            return self._fixed[0].copy()
        keys = self._unbits(np.asarray(self._map.key_array()))
        if not self._specials:
            return keys
        out = np.zeros(len(keys) + len(self._specials), dtype=self._np)
        out[self._public(np.arange(len(keys)))] = keys
        if self.nan_count > 0 and self._is_float:
            out[self.nan_index] = np.nan
        return out

    def keys(self):
        out = self.key_array().tolist()
        if self.has_null:
            out[self.null_index] = None
        return out

    def map_ordinal(self, values):
        """ordinals of `values` (-1: unknown key), as the narrowest signed integer type that holds len(self)
        (src/hash_primitives.hpp:611-691); masked inputs are the caller's business (vaex/hash.py:203-211)"""
        values = np.ma.getdata(values) if np.ma.isMaskedArray(values) else np.asarray(values)
        bits, nan = self._bits(values)
        ords = np.asarray(self._map.map_ordinal(bits)) if len(bits) else np.zeros(0, dtype=np.int64)
        if self._perm is not None:
            ords = np.where(ords >= 0, self._perm[np.maximum(ords, 0)], -1)
        elif self._fixed is None:
            ords = self._public(ords)
        if nan is not None:
            ords = np.where(nan, self.nan_index if self.nan_count > 0 else -1, ords)
        return ords.astype(_narrowest(len(self)))

    def isin(self, values):
        # src/hash_primitives.hpp:539-565: a NaN is in the set when the set saw one
        return np.asarray(self.map_ordinal(values)).astype(np.int64) >= 0

    def flatten_values(self, values, map_index, out):
        out[...] = values  # one table: local ordinals are global ordinals (src/hash_primitives.hpp:540-565 adds shard offsets)
        return out

    def offsets(self):
        return [0]

    @property
    def offset(self):
        return 0

    def extract(self):
        keys = self.key_array()
        return {k: i for i, k in enumerate(keys.tolist()) if i != self.null_index}

    def bytes_used(self):
        """bytes of the keys held (src/hash_primitives.hpp:64-71 counts (key, value) pairs the same way)"""
        return int(len(self)) * 16

    def __sizeof__(self):
        return self.bytes_used()

    def __reduce__(self):
        # vaex/hash.py:21-25 pickles sets as (type, (keys, null_index, nan_count, null_count, fingerprint))
        return (type(self), (self.key_array(), self.null_index, self.nan_count, self.null_count, self.fingerprint))


def _make(dtype_name):
    return type("ordered_set_" + dtype_name, (_OrderedSet,), {"dtype_name": dtype_name, "__doc__": f"GPU-backed vaex.superutils.ordered_set_{dtype_name}"})


CLASSES = {name: _make(name) for name in _INT + _FLOAT}
for _name, _cls in CLASSES.items():
    globals()["ordered_set_" + _name] = _cls
