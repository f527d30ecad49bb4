"""Test-infrastructure stand-in for `blake3` (fingerprints only; any stable hash works)."""
import hashlib

__version__ = "0.3.0"


class blake3:
    AUTO = -1

    def __init__(self, data=None, max_threads=1, **kw):
        self._h = hashlib.blake2b(digest_size=32)
        if data is not None:
            self._h.update(data)

    def update(self, data):
        self._h.update(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data)
        return self

    def hexdigest(self):
        return self._h.hexdigest()

    def digest(self):
        return self._h.digest()
