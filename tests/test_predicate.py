"""vaex_amd.predicate: selection expression strings -> (columns, comparison terms, truth table) of the C-ABI's device-side
selections (include/vaex_hip.h); the numpy evaluation of the compiled form must agree with numpy evaluating the expression
itself, NaN and all (the reference's selection semantics: vaex/execution.py:530-549 evaluates the same string with numpy)."""
import numpy as np
import pytest

from vaex_amd import predicate as P

rng = np.random.default_rng(5)
N = 5000
COLS = {
    "x": np.where(rng.random(N) < 0.05, np.nan, rng.normal(0, 2, N)),
    "y": rng.normal(0, 1, N).astype("f4"),
    "i": rng.integers(-50, 50, N),
    "u": rng.integers(0, 200, N).astype("u1"),
}

CASES = [  # ("x != x" style column-column comparisons are outside the subset: see the refusal test)
    "x > 0", "x >= 0.5", "x < -1", "x <= 0", "x == 0", "x != 0", "1 < x", "(-1.5 <= x) & (x < 2)", "i == 3", "i != -7", "u >= 128",
    "(x > 0) & (y < 0.25)", "(x > 0) | (i < 0)", "~(x > 0)", "~((x > 0) & (i >= 10)) | (u == 7)", "(x > 0) & (y > 0) & (i > 0) & (u > 100)",
    "(0 < x) & (x < 1) | (i == 0)", "x > 1e-3", "i > 2.5",
]


@pytest.mark.parametrize("expr", CASES)
def test_compiled_form_equals_numpy(expr):
    p = P.compile_selection(expr, COLS)
    with np.errstate(invalid="ignore"):
        want = eval(expr, {}, dict(COLS))
    assert np.array_equal(p.numpy_mask(COLS), want), expr
    assert len(p.terms) <= P.MAX_TERMS and len(p.columns) <= P.MAX_COLUMNS and 0 <= p.truth < (1 << (1 << len(p.terms)))


# (chained comparisons, `and` / `or` / `not`: vaex keeps the last link / operand or refuses — left to vaex, see vaex_amd/predicate.py)
@pytest.mark.parametrize("expr", ["-1.5 <= x < 2", "x > 0 and i < 3", "x > 0 or i < 3", "not (x > 0)", "i + 1 > 0", "x > y", "sin(x) > 1", "x ** 3 > 1", "x ** 0.5 > 1", "z > 0", "x > 0 & y < 1", "(x>0)&(x>1)&(x>2)&(x>3)&(x>4)", "x", "x > 'a'", "x >", "i > 99999999999999999999"])
def test_everything_else_is_refused(expr):
    with pytest.raises(P.Unsupported):
        P.compile_selection(expr, COLS)


def test_identical_comparisons_share_a_term_and_keys_identify_predicates():
    a = P.compile_selection("(x > 0) & (x > 0) & (i < 3)", COLS)
    b = P.compile_selection("(i < 3) & (x > 0)", COLS)
    assert len(a.terms) == 2
    assert a.key() != b.key()  # (different column order: different objects, same rows)
    assert np.array_equal(a.numpy_mask(COLS), b.numpy_mask(COLS))


# round 5: arithmetic over float64 columns (+ - * / unary minus, ** 2, sqrt, abs, numbers) and virtual columns are compiled into postfix
# programs (vxh_selection_set_program); what numpy computes for these operations is correctly rounded, so the host evaluation below is
# also what the device must produce
ARITH = [("2*x + 1 > 0", None), ("x**2 + y**2 < 4", None), ("(r < 1.5) & (x > -1)", {"r": "sqrt(x**2 + y**2)"}), ("abs(x - y)/2 >= 0.25", None),
         ("-x < 0.5", None), ("3 > x*y", None), ("(x / y > 2) | (1 / x < -3)", None), ("r2 - 1 != 0", {"r2": "x*x", "unused": "sin(x)"}),
         ("((x + y) * (x - y)) / (1 + x**2) <= 0.1", None),
         # round 6: <expression> <op> <expression> — both sides and the comparison in one program (SEL_CMP), the term is `<program> != 0`
         ("x > y", None), ("x != x", None), ("(x <= y) & (x > -1)", None), ("x + y <= 2 * x", None), ("abs(x) >= y**2", None), ("~(x == y) | (r < x)", {"r": "sqrt(x**2 + y**2)"}),
         ("x - y < y / x", None)]


@pytest.mark.parametrize("expr,virtual", ARITH)
def test_arithmetic_terms_equal_numpy(expr, virtual):
    cols = dict(COLS)
    cols["y"] = np.linspace(-2.0, 2.0, len(cols["x"]))
    p = P.compile_selection(expr, cols, virtual=virtual)
    ns = dict(cols, sqrt=np.sqrt, abs=np.abs)
    for k, v in (virtual or {}).items():
        if k in expr:
            ns[k] = eval(v, {}, ns)
    with np.errstate(all="ignore"):
        want = eval(expr, {}, ns)
    assert np.array_equal(p.numpy_mask(cols), want), expr
    assert p.programs and all(len(st) <= P.MAX_STEPS for st in p.programs.values())
    assert p.key() != P.compile_selection("x > 0", cols).key()


def test_arithmetic_is_float64_only_and_bounded():
    cols = dict(COLS)
    cols["f"] = cols["x"].astype("f4")
    for expr in ("f * 2 > 1", "i + 1 > 2", "x + i > 0", "x ** 3 > 1", "exp(x) > 1", "x > f", "i > x", "1 > 2 + 3"):   # (x > x * 2 is a program since round 6; other dtypes on either side still decline)
        with pytest.raises(P.Unsupported):
            P.compile_selection(expr, cols)
    with pytest.raises(P.Unsupported):   # more than sixteen steps
        P.compile_selection(" + ".join(["x * x"] * 9) + " > 1", cols)
    # a virtual column that is just another name of a real column is a plain term
    assert not P.compile_selection("alias > 1", cols, virtual={"alias": "x"}).programs


def test_random_expressions_keep_the_rows_numpy_keeps():
    """a differential fuzz of the subset's host evaluation (Predicate.numpy_mask — the specification the device kernels are tested against,
    tests/test_gpu_selection.py) against plain numpy on the same expression string — which is what vaex evaluates (vaex/scopes.py:138-177; 3000
    such expressions were also run against the real package's df.evaluate while this test was written: 0 differences)"""
    from tests.predicate_fuzz import random_expression
    rng0 = np.random.default_rng(5)
    n = 3000
    x = rng0.normal(0, 2, n); x[::13] = np.nan; x[5] = np.inf; x[6] = -np.inf; x[7] = -0.0
    f = rng0.choice(np.array([0.1, 0.3, 0.30000001, 0.5, -0.7, 1e-8], dtype="f4"), n); f[::17] = np.nan
    i = rng0.integers(-5, 6, n).astype("i8"); i[3] = 2**62; i[4] = -2**62
    cols = dict(x=x, y=rng0.normal(1, 1, n), f=f, i=i, i4=rng0.integers(-100, 100, n).astype("i4"), h=rng0.integers(-300, 300, n).astype("i2"),
                u1=rng0.integers(0, 255, n).astype("u1"), u4=rng0.integers(0, 2**32 - 1, n, dtype="u8").astype("u4"), b=rng0.integers(0, 2, n).astype(bool))
    scope = dict(cols, sqrt=np.sqrt, abs=np.abs)
    inside = 0
    for seed in range(600):
        expr = random_expression(np.random.default_rng(seed), list(cols), ["x", "y"])
        try:
            pred = P.compile_selection(expr, cols)
        except P.Unsupported:
            continue
        inside += 1
        with np.errstate(all="ignore"):
            want = np.asarray(eval(expr, {}, scope)).astype(bool)
        assert np.array_equal(pred.numpy_mask(cols), want), expr
    assert inside > 400, inside


def test_constant_subtrees_are_folded_with_python_semantics():
    """vaex evaluates a selection with Python's eval(): a subexpression made of numbers only is folded in exact integer arithmetic before
    anything becomes a float64 — `(2**53 + 1 + 1) * x` multiplies by ...994, the step-by-step float64 folding the device used to do gives
    ...992."""
    cols = {"x": np.array([1.0, 1.0000000000000002, 0.9999999999999999, -1.0, np.nan])}
    p = P.compile_selection("(9007199254740992 + 1 + 1) * x > 9007199254740994", cols)
    consts = [v for op, _, v in p.programs[0] if op == P.SEL_CONST]
    assert consts == [9007199254740994.0]
    with np.errstate(invalid="ignore"):
        want = eval("(9007199254740992 + 1 + 1) * x > 9007199254740994", {}, dict(cols))
    assert np.array_equal(p.numpy_mask(cols), want) and want.tolist() == [False, True, False, False, False]
    # the constant side of a comparison is folded too, and stays an int where Python's is one
    q = P.compile_selection("x > 2 * 3 - 5", cols)
    assert not q.programs and q.terms == [(0, 2, 1)] and isinstance(q.terms[0][2], int)
    r = P.compile_selection("x * (1 / 3) >= 1 / 3", cols)
    assert [v for op, _, v in r.programs[0] if op == P.SEL_CONST] == [1 / 3] and r.terms[0][2] == 1 / 3
    assert P.compile_selection("x < 2 ** -1", cols).terms == [(0, 0, 0.5)]
    for expr in ("x > 1 / 0", "x * 10 ** 400 > 1", "x > 2 ** 1000 ** 1000", "x > (-8) ** 0.5"):
        with pytest.raises(P.Unsupported):
            P.compile_selection(expr, cols)


@pytest.mark.parametrize("expr,virtual", [(e, None) for e in CASES] + ARITH)
def test_torch_mask_equals_numpy_mask(expr, virtual):
    """Predicate.torch_mask — what Frame's ready-made keep-masks over DEVICE columns are built with (minmax, percentiles, the hashed
    groupby) — keeps numpy_mask's rows, arithmetic programs included (the programs used to be dropped on that road).  Here on
    CPU tensors; tests/test_gpu_selection.py runs the same on the GPU."""
    torch = pytest.importorskip("torch")
    cols = dict(COLS)
    if virtual is not None or expr in [a for a, _ in ARITH]:
        cols["y"] = np.linspace(-2.0, 2.0, len(cols["x"]))
    p = P.compile_selection(expr, cols, virtual=virtual)
    tensors = {c: torch.from_numpy(np.ascontiguousarray(cols[c])) for c in p.columns}
    with np.errstate(all="ignore"):
        want = p.numpy_mask({c: cols[c] for c in p.columns})
    got = p.torch_mask(tensors)
    assert got.dtype == torch.uint8 and np.array_equal(got.numpy().astype(bool), want), expr
