"""CPU-side checks of the drop-in boundary: libvaexhip.so loads and exports every symbol
include/vaex_hip.h declares; the pybind11 shim exposes the reference's class surface (names,
constructors, properties, sizes, errors); nothing computes without a GPU (no CPU fallback)."""
import ctypes
import os
import pickle
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DTYPES = ["float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool"]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vaex_hip.h")).read()
    names = sorted(set(re.findall(r"\b(vxh_[a-z0-9_]+)\s*\(", header)))
    assert len(names) > 40
    import vaex_amd  # noqa: F401  (torch first, then the library: one HIP runtime per process)
    lib = ctypes.CDLL(os.path.join(ROOT, "vaex_amd", "lib", "libvaexhip.so"))
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.vxh_abi_version.restype = ctypes.c_int
    assert lib.vxh_abi_version() == 1


def test_class_surface_names(sa):
    # names vaex looks up: find_type_from_dtype(vaex.superagg, prefix, dtype) (vaex/utils.py:754-791)
    for dt in DTYPES:
        for nn in ("", "_non_native"):
            for prefix in ("BinnerScalar_", "BinnerOrdinal_", "AggCount_", "AggSum_", "AggSumMoment_", "AggMin_", "AggMax_"):
                assert hasattr(sa, prefix + dt + nn), prefix + dt + nn
    for name in ("Grid", "Binner", "Aggregator"):
        assert hasattr(sa, name)
    # the rest of the aggregator family resolves on demand (module __getattr__): src/agg_first.cpp:165-178, agg_nunique.cpp:226-235, agg_list.cpp:223-233
    for dt in DTYPES:
        for nn in ("", "_non_native"):
            assert callable(getattr(sa, "AggNUnique_" + dt + nn))
            assert callable(getattr(sa, "AggList_" + dt + "_int64" + nn))
            assert callable(getattr(sa, "AggFirst_" + dt + "_float64" + nn))
    for missing in ("AggCount_string", "AggNUnique_string", "BinnerCombined"):
        assert not hasattr(sa, missing)


def test_collector_surface_without_a_gpu(sa):
    import sys
    g0 = sa.Grid([])
    b = sa.BinnerScalar_float64(4, "x", 0.0, 1.0, 10)
    g = sa.Grid([b])
    # vaex predicts nunique's footprint as sizeof(class on an empty grid) x cells and insists on equality (vaex/agg.py:354-368)
    one = sys.getsizeof(sa.AggNUnique_float64(g0, 1, 4, False, True))
    assert sys.getsizeof(sa.AggNUnique_float64(g, 1, 4, False, True)) == one * len(g)
    with pytest.raises(RuntimeError, match="Expected 1 grid"):       # src/agg_nunique.cpp:20
        sa.AggNUnique_int32(g, 2, 4, False, False)
    with pytest.raises(RuntimeError, match="only accepts 1 grid"):   # src/agg_list.cpp:18
        sa.AggList_float64_int64(g, 2, 4, False, False)
    a = sa.AggNUnique_int32(g, 1, 4, True, False)
    with pytest.raises(RuntimeError, match="Itemsize"):
        a.set_data(0, __import__("numpy").zeros(4, dtype="f8"), 0)
    with pytest.raises(RuntimeError, match="merge not implemented"):  # src/agg_nunique.cpp:46-49
        a.merge([a])
    sa.AggList_float64_int64(g, 1, 4, False, False).merge([])         # (a no-op in the reference: src/agg_list.cpp:49)


def test_binner_scalar_surface(sa):
    b = sa.BinnerScalar_float64(4, "x", 0.0, 5.0, 5)
    assert len(b) == 8 and b.expression == "x" and b.bins == 5 and b.vmin == 0.0 and b.vmax == 5.0
    assert "BinnerScalar_" in str(b)  # vaex/agg.py:327
    c = b.copy()
    assert type(c) is type(b) and c.vmax == 5.0 and c is not b
    p = pickle.loads(pickle.dumps(b))
    assert type(p) is type(b) and (p.bins, p.vmin, p.vmax, p.expression) == (5, 0.0, 5.0, "x")
    with pytest.raises(RuntimeError, match="Expected a 1d array"):
        b.set_data(0, np.zeros((2, 2)))
    with pytest.raises(RuntimeError, match="Itemsize of data and binner are not equal"):
        b.set_data(0, np.zeros(4, dtype="f4"))
    with pytest.raises(RuntimeError, match="thread out of bound"):
        b.set_data(4, np.zeros(4))
    b.set_data(3, np.zeros(4)); b.set_data_mask(3, np.zeros(4, dtype=bool)); b.clear_data_mask(3)


def test_binner_ordinal_surface(sa):
    b = sa.BinnerOrdinal_int32(2, "k", 10, 3, False, False)
    assert len(b) == 12 and b.ordinal_count == 10 and b.min_value == 3 and b.allow_other is False
    assert "BinnerOrdinal_" in str(b)
    assert len(sa.BinnerOrdinal_int32(2, "k", 10, 3, True, False)) == 13
    assert type(b.copy()) is type(b)
    assert pickle.loads(pickle.dumps(b)).min_value == 3


def test_grid_shapes_strides(sa):
    bx = sa.BinnerScalar_float64(1, "x", 0, 1, 128)
    by = sa.BinnerScalar_float32(1, "y", 0, 1, 256)
    bk = sa.BinnerOrdinal_int8(1, "k", 5, 0, False, False)
    g = sa.Grid([bx, by, bk])
    assert g.shapes == [131, 259, 7] and g.strides == [1, 131, 131 * 259] and len(g) == 131 * 259 * 7
    assert g.binners[0] is bx and g.binners[2] is bk
    assert len(sa.Grid([])) == 1


@pytest.mark.parametrize("cls,dtype,cell", [("AggCount_", "float32", "int64"), ("AggSum_", "float32", "float64"), ("AggSum_", "int8", "int64"), ("AggSum_", "uint16", "uint64"),
                                            ("AggSum_", "bool", "int64"), ("AggMin_", "int8", "int8"), ("AggMax_", "float32", "float32"), ("AggMax_", "bool", "bool"), ("AggMin_", "uint64", "uint64")])
def test_aggregator_sizes_and_buffers(sa, cls, dtype, cell):
    g = sa.Grid([sa.BinnerScalar_float64(3, "x", 0, 1, 4), sa.BinnerOrdinal_int64(3, "k", 3, 0, False, False)])
    grids = 2
    a = getattr(sa, cls + dtype)(g, grids, 3)
    # size contract: vaex/agg.py:311-318 raises unless sys.getsizeof == itemsize * cells * grids
    assert sys.getsizeof(a) == np.dtype(cell).itemsize * 7 * 5 * grids
    assert a.grid is g
    buf = np.asarray(a)
    assert buf.dtype == np.dtype(cell) and buf.shape == (grids, 7, 5)
    assert buf.strides == (35 * buf.itemsize, buf.itemsize, 7 * buf.itemsize)  # dim 0 fastest: agg_base.hpp:106-125
    r = a.get_result()
    assert r.shape == (7, 5) and r.dtype == np.dtype(cell)
    if cls == "AggMin_":
        assert np.all(r == (np.iinfo(cell).max if np.dtype(cell).kind in "iu" else np.inf))
    elif cls == "AggMax_":
        want = {"float32": -np.inf, "bool": False}[dtype]
        assert np.all(r == want)
    else:
        assert np.all(r == 0)
    # host-side seeding + merge work without a device
    buf[0, 1, 2] = 1
    b = getattr(sa, cls + dtype)(g, grids, 3)
    a.merge([b])
    assert a.get_result()[1, 2] == 1


def test_sum_moment_ctor(sa):
    g = sa.Grid([sa.BinnerScalar_float64(1, "x", 0, 1, 4)])
    a = sa.AggSumMoment_float64(g, 1, 1, 2)
    assert sys.getsizeof(a) == 8 * 7


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_fails_loudly(sa):
    assert sa.device_count() == 0
    b = sa.BinnerScalar_float64(1, "x", 0, 1, 4)
    g = sa.Grid([b])
    a = sa.AggCount_int64(g, 1, 1)
    b.set_data(0, np.zeros(10)); b.clear_data_mask(0); a.clear_data_mask(0)
    with pytest.raises(RuntimeError, match="no HIP device"):
        g.bin(0, [a], 10)
    with pytest.raises(RuntimeError, match="no HIP device"):
        sa.ordered_set_int64()


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vaex_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), os.path.join(dirpath, f)


def test_heavy_key_sample_finds_the_head_of_a_zipf_law_and_nothing_in_uniform_keys():
    """Frame._heavy_keys (host logic of the groupby's heavy-hitter peel): keys with >= 1/128 of a strided sample, ascending, remembered
    per column object; uniform keys give None"""
    import numpy as np
    from vaex_amd.binned import Frame
    rng = np.random.default_rng(1)
    n = 3_000_000
    z = rng.zipf(1.3, n)
    k = (np.minimum(z, 400_000) * 2654435761) % (1 << 40)
    f = Frame.__new__(Frame)
    f.n = n
    heavy = f._heavy_keys("k", k)
    share = {key: float((k == key).mean()) for key in heavy}
    assert heavy is not None and 8 <= len(heavy) <= 128 and np.all(np.diff(heavy) > 0)
    assert all(s > 0.004 for s in share.values()), share                       # nothing light is called heavy
    top = [(c * 2654435761) % (1 << 40) for c in (1, 2, 3, 400_000)]            # the head of the law, and the clipped "default" value
    assert all(t in set(heavy.tolist()) for t in top)
    assert f._heavy_keys("k", k) is heavy                                      # remembered per column object
    g = Frame.__new__(Frame)
    g.n = n
    assert g._heavy_keys("k", (rng.integers(0, 1_000_000, n) * 2654435761) % (1 << 40)) is None


def test_heavy_key_sampler_host_logic():
    """round 4: the keys the fused groupby peels inside its scatter kernel come from a strided sample of the key column — those holding
    >= `share` of it, the 128 most frequent at most, ascending; remembered per (column, share).  Host logic only (numpy keys): no GPU"""
    import numpy as np
    from vaex_amd.binned import Frame
    rng = np.random.default_rng(0)
    n = 1 << 22
    k = rng.integers(0, 1_000_000, n).astype(np.int64)
    k[rng.random(n) < 0.2] = 777
    k[rng.random(n) < 0.01] = -5
    f = Frame(dict(k=k, v=rng.normal(size=n)))
    assert list(f._heavy_keys("k", f.columns["k"], share=1 / 1024)) == [-5, 777]
    assert list(f._heavy_keys("k", f.columns["k"])) == [-5, 777]            # (1 / 128: both are above it too)
    assert f._heavy_keys("k", f.columns["k"], share=1 / 1024) is f._heavy_keys("k", f.columns["k"], share=1 / 1024)   # remembered
    # more than 128 keys above the threshold: the 128 most frequent
    many = np.repeat(np.arange(300, dtype=np.int64), np.r_[np.full(150, 40_000), np.full(150, 20_000)])
    rng.shuffle(many)
    g = Frame(dict(k=many, v=np.zeros(len(many))))
    top = g._heavy_keys("k", g.columns["k"], share=1 / 1024)
    assert len(top) == 128 and set(top) <= set(range(150))
    # uniform keys: none
    u = Frame(dict(k=rng.integers(0, 1_000_000, n).astype(np.int64), v=np.zeros(n)))
    assert u._heavy_keys("k", u.columns["k"], share=1 / 1024) is None


def test_the_product_library_ships_no_ablation_switches():
    """timing experiments that make results wrong on purpose are not reachable through the product's C-ABI.
    vxh_config_set takes `no_pipeline` bits 1 / 16 only (A/B switches to the generic kernels, results unchanged), refuses every other
    bit and `gb_abl` by name, and the knobs that lost their A/B are gone; the sources keep such branches behind VXH_ABL / VXH_GB_ABL,
    compile-time zeros unless the ablation build (`make ablate` -> tools/ablate/) defines VXH_ABLATE.  No device needed."""
    import re
    import pytest
    import vaex_amd
    sa = vaex_amd.superagg
    for ok in (0, 1, 16, 17, 0):
        sa.config_set("no_pipeline", ok)
        assert sa.config_get("no_pipeline") == ok
    for bits in (2, 64, 128, 256, 512, 1024, 2048, 8192, 64 | 1):
        with pytest.raises(RuntimeError, match="ablation build"):
            sa.config_set("no_pipeline", bits)
    assert sa.config_get("no_pipeline") == 0
    with pytest.raises(RuntimeError, match="ablation build"):
        sa.config_set("gb_abl", 1)
    for gone in ("merge_fused", "gb_known_count", "f64_rec12"):
        with pytest.raises(RuntimeError, match="unknown config key"):
            sa.config_set(gone, 1)
    csrc = os.path.join(ROOT, "vaex_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".hpp", ".cpp")):
            src = open(os.path.join(csrc, f)).read()
            assert "wrong on purpose" not in src, f
            # a kernel may look at the experiment bits only through the macros (bits 1 and 16 are the launchers' strategy switches)
            for m in re.finditer(r"no_pipeline & (\d+)", src):
                assert int(m.group(1)) in (1, 16, 17), (f, m.group(0))
            assert not re.search(r"[GP]\.abl &", src), f
