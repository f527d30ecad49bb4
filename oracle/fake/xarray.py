"""TEST INFRASTRUCTURE — the few members of xarray.DataArray that vaex's binby / count(array_type='xarray') build and the reference's
tests read (vaex/groupby.py:838-862, vaex/dataframe.py:907-923; tests/groupby_test.py:178-216, :606-643): xarray is not in this image."""
import numpy as np


class _Coord:
    def __init__(self, values):
        self.values = np.asarray(values)
        self.data = self.values

    def tolist(self):
        return self.values.tolist()


class DataArray:
    def __init__(self, data, coords=None, dims=None):
        self.data = np.asanyarray(data)
        self.values = self.data
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self.data.ndim))
        if coords is None:
            coords = [np.arange(n) for n in self.data.shape]
        if isinstance(coords, dict):
            self.coords = {k: _Coord(v) for k, v in coords.items()}
        else:
            coords = list(coords)
            assert len(coords) == len(self.dims), (len(coords), self.dims)
            for c, n in zip(coords, self.data.shape):
                assert len(c) == n, "coordinate length does not match the dimension"
            self.coords = {d: _Coord(c) for d, c in zip(self.dims, coords)}

    @property
    def shape(self):
        return self.data.shape

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.data, dtype=dtype)

    def tolist(self):
        return self.data.tolist()
