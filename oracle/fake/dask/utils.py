def parse_bytes(s):
    if isinstance(s, (int, float)):
        return int(s)
    s = s.strip().lower().replace(" ", "")
    units = {"kb": 10**3, "mb": 10**6, "gb": 10**9, "tb": 10**12, "kib": 2**10, "mib": 2**20, "gib": 2**30, "tib": 2**40, "b": 1, "k": 10**3, "m": 10**6, "g": 10**9}
    for u in sorted(units, key=len, reverse=True):
        if s.endswith(u):
            return int(float(s[: -len(u)] or 1) * units[u])
    return int(float(s))
