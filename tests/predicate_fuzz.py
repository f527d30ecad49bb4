"""Random selection expressions for the differential tests of the predicate subset (tests/test_predicate.py on the host,
tests/test_gpu_selection.py on the device): comparisons of any column with integer / float / huge / float32-boundary constants (either
side), arithmetic over float64 columns, combined with & | ~ three levels deep.  What comes out is NOT always inside the subset
(too many terms, ...): callers skip what compile_selection refuses."""
OPS = ["<", "<=", ">", ">=", "==", "!="]


def _const(rng):
    kind = rng.choice(["int", "float", "neg", "big", "frac"])
    if kind == "int":
        return str(int(rng.integers(-6, 7)))
    if kind == "float":
        return repr(float(round(float(rng.normal(0, 2)), 3)))
    if kind == "neg":
        return "-" + repr(float(abs(round(float(rng.normal(0, 2)), 2))))
    if kind == "big":
        return str(rng.choice(["2147483648", "4294967296", "9007199254740993", "-1099511627776", "255", "256", "4294967295", "1e300", "-1e-300", "65535", "-129"]))
    return str(rng.choice(["0.3", "0.1", "0.30000001", "0.5", "1e-8", "2.5", "-0.0", "0.0"]))


def _term(rng, names, f64):
    name = str(rng.choice(names))
    c = _const(rng)
    if rng.random() < 0.2:
        return f"({c} {rng.choice(OPS)} {name})"
    if rng.random() < 0.25 and f64:
        a, b = str(rng.choice(f64)), str(rng.choice(f64))
        e = str(rng.choice([f"{a} + {b}", f"{a} * 2 - {b}", f"{a}**2", f"abs({a})", f"sqrt({b}**2)", f"-{a}", f"{a}/{b}", f"({a} - {b})/2 + 1"]))
        return f"({e} {rng.choice(OPS)} {c})"
    if rng.random() < 0.15 and len(f64) >= 2:   # round 6: column / expression against column / expression
        a, b = str(rng.choice(f64)), str(rng.choice(f64))
        return str(rng.choice([f"({a} {rng.choice(OPS)} {b})", f"({a} + 1 {rng.choice(OPS)} {b} * 2)", f"(abs({a}) {rng.choice(OPS)} {b}**2)", f"({a} - {b} {rng.choice(OPS)} {b} / {a})"]))
    return f"({name} {rng.choice(OPS)} {c})"


def random_expression(rng, names, f64, depth=0):
    r = rng.random()
    if depth > 2 or r < 0.4:
        return _term(rng, names, f64)
    if r < 0.5:
        return f"~{random_expression(rng, names, f64, depth + 1)}"
    return f"({random_expression(rng, names, f64, depth + 1)} {rng.choice(['&', '|'])} {random_expression(rng, names, f64, depth + 1)})"
