class frozendict(dict):
    def __hash__(self):
        return hash(tuple(sorted(self.items(), key=lambda kv: repr(kv[0]))))

    def _ro(self, *a, **k):
        raise TypeError("frozendict is immutable")

    __setitem__ = __delitem__ = clear = pop = popitem = setdefault = update = _ro
