"""-m gpu: seeded random cases over the signature space of the partition / LDS kernels (dims, shapes, value columns,
masks, NaNs, ragged row counts, strategies, second-generation pass 1 on/off, forced hot boxes) against the oracle."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests import cases

pytestmark = pytest.mark.gpu

WV_DEFAULT = 6  # pass 1 next to a hot box: part_scatter_wv, cold records in slab-sorted groups written in chip-wide bursts (5: as they come; 3: without rings, one record stream per (wave, slab))
KEYS = ("strategy", "wv", "wv_waves", "wv_waves_direct", "wv_block", "blk", "hot", "hot_min_rows", "hot_min_pct", "hot_x0", "hot_y0", "hot_w", "hot_h", "part_chunk", "count16", "wv_phase")


def _reset(sa):
    for k in KEYS:
        sa.config_set(k, {"blk": 1, "hot": 1, "count16": 1, "wv": WV_DEFAULT, "wv_waves": 8, "wv_waves_direct": 16, "wv_phase": 12}.get(k, 0))


#: 400 seeds in the suite (160 until round 5); a soak run sets VAEX_AMD_FUZZ_SEEDS (profiles/r05_fuzz_soak.txt: 3000 seeds)
@pytest.mark.parametrize("seed", range(int(os.environ.get("VAEX_AMD_FUZZ_SEEDS", "400"))))
def test_fuzz_against_oracle(sa, gpu_ready, seed):
    rng = np.random.default_rng(1000 + seed)
    ndim = int(rng.integers(1, 4))
    n = int(rng.choice([1, 777, 4095, 4097, 50_000, 131_072, 300_001]))
    shape = {1: [int(rng.choice([64, 5000, 300_000]))], 2: [int(rng.choice([16, 200, 700]))] * 2, 3: [int(rng.choice([8, 48, 100]))] * 3}[ndim]
    if ndim == 2 and rng.random() < 0.5:
        shape[1] = int(rng.choice([16, 300]))
    cols = [rng.normal(rng.uniform(-1, 1), rng.uniform(0.3, 2.0), n) for _ in range(ndim)]
    for c in cols:
        c[rng.random(n) < 0.01] = np.nan
    v = rng.normal(3, 2, n)
    v[rng.random(n) < 0.02] = np.nan
    w = rng.normal(0, 1, n)
    m = rng.random(n) < 0.6
    binners = [dict(kind="scalar", data=c, vmin=-4, vmax=4, bins=s) for c, s in zip(cols, shape)]
    menu = [
        [dict(kind="count")],
        [dict(kind="count"), dict(kind="sum", data=v), dict(kind="count", data=v)],
        [dict(kind="sum", data=v)],
        [dict(kind="count", mask=m)],
        [dict(kind="sum", data=v, mask=m), dict(kind="count", data=v, mask=m)],
        [dict(kind="count"), dict(kind="max", data=v), dict(kind="summoment", data=v, moment=2)],
        [dict(kind="sum", data=v), dict(kind="sum", data=w), dict(kind="count", mask=m)],
    ]
    aggs = menu[int(rng.integers(0, len(menu)))]
    case = dict(n=n, binners=binners, aggs=aggs)
    want = oracle.run_case(case)
    try:
        sa.config_set("strategy", int(rng.choice([0, 0, 4, 4, 3])))
        sa.config_set("blk", int(rng.choice([1, 2, 0])))
        # third-generation pass 1 (part_scatter_wv): off / auto / also next to a hot box / there without rings, one record
        # stream per (wave, slab) / per (workgroup, slab) / slab-sorted groups in one stream per wave / ... held back for chip-wide bursts
        wv = int(rng.integers(0, 2)) * (1 + seed % 6)
        sa.config_set("wv", wv)
        sa.config_set("wv_waves_direct", [16, 8, 4, 12][seed % 4])
        sa.config_set("wv_waves", [4, 6, 8, 12, 16][seed % 5])
        sa.config_set("wv_phase", [13, 4, 9, 12][seed % 4])   # (the write bursts' wall-clock bit: 4 = a flip every 160 ns)
        sa.config_set("wv_block", [0, 64, 320][seed % 3])  # tiny queue blocks: many blocks per (wave, slab), in-line reservations
        sa.config_set("count16", int(rng.choice([1, 2])))
        if rng.random() < 0.5:
            sa.config_set("part_chunk", 1 << 20)
        if ndim == 2 and rng.random() < 0.7:
            sx, sy = shape[0] + 3, shape[1] + 3
            bw, bh = int(rng.integers(1, min(sx, 90) + 1)), int(rng.integers(1, min(sy, 90) + 1))
            for k, val in zip(("hot_x0", "hot_y0", "hot_w", "hot_h"), (int(rng.integers(0, sx - bw + 1)), int(rng.integers(0, sy - bh + 1)), bw, bh)):
                sa.config_set(k, val)
        elif rng.random() < 0.5:
            sa.config_set("hot_min_rows", 1)
            sa.config_set("hot_min_pct", 5)
        got = cases.run_superagg(sa, case)
        cases.assert_case_equal(got, want, case)
    finally:
        _reset(sa)
