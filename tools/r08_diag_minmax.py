"""Diagnosis (round 6 soak, random call 5332): a delayed aggregation with limits='minmax' next to another delayed aggregation — execute() raised
TypeError: unhashable type: 'numpy.ndarray' from BinnerScalar.__hash__ under install().  Prints what the binner's limits are on both sides."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "_ref", "vaexpy"), os.path.join(ROOT, "oracle", "fake"), ROOT]
import numpy as np
import vaex, vaex.dataframe
import vaex_amd

n = 120_000
r = np.random.default_rng(21)
x = r.normal(0, 1, n); x[::997] = np.nan
df = vaex.from_arrays(x=x, y=r.normal(0, 1, n), u1=r.integers(0, 200, n).astype("u1"), b=r.random(n) < 0.3, v=r.normal(3, 2, n),
                      m=np.ma.array(r.normal(0, 1, n), mask=r.random(n) < 0.05))
orig_hash = vaex.dataframe.BinnerScalar.__hash__
def loud_hash(self):
    try:
        return orig_hash(self)
    except TypeError:
        print("   unhashable binner:", self.expression, type(self.minimum), getattr(self.minimum, "shape", None), getattr(self.minimum, "dtype", None), repr(self.minimum), repr(self.maximum), flush=True)
        raise
vaex.dataframe.BinnerScalar.__hash__ = loud_hash
sel = "((((1.682 != y) | (y/y >= 2147483648)) | (-0.463 == b)) | (((y * 2 - x <= 65535) | (-2.13 > x)) & (v >= 0.126)))"
for side in ("reference", "install"):
    if side == "install":
        vaex_amd.install()
    for frame in ("plain", "sliced"):
        d = df if frame == "plain" else df[3000:n - 10_000]
        for selection in (None, sel):
            for other in (False, True):
                try:
                    mm = d.minmax("u1", selection=selection)
                    p = d.sum("m", binby=["u1"], limits="minmax", shape=[29], selection=selection, delay=True)
                    q = d.count(binby=["x"], limits=[[-3, 3]], shape=[5], delay=True) if other else None
                    d.execute()
                    res = np.asarray(p.get())
                    print(side, frame, "sel" if selection else "-", "with another task" if other else "alone", "ok", type(mm), getattr(mm, "shape", None), getattr(mm, "dtype", None), float(np.nansum(res)))
                except Exception as e:
                    print(side, frame, "sel" if selection else "-", "with another task" if other else "alone", "RAISED", type(e).__name__, str(e)[:100])
print("DONE")
