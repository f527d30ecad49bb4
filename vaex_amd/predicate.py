"""Selection expressions -> device-side predicates (include/vaex_hip.h "device-side selections").

vaex keeps a selection as a boolean expression string (vaex/selections.py:50-76 SelectionExpression.boolean_expression,
validated against the AST whitelist of vaex/expresso.py:46-155) and evaluates it with numpy on every chunk
(vaex/execution.py:530-549).  `compile_selection` accepts the subset the GPU evaluates itself:

    comparisons  <column> (< <= > >= == !=) <number>,  <number> (op) <column>
    combined with  &  |  ~,  at most 4 comparisons over at most 4 columns

Left to vaex on purpose: `and` / `or` and chained comparisons (`a < x <= b`) — vaex's expression rewriter keeps only the LAST operand
/ link of those (vaex/expresso.py:438-446 visit_Compare and its BoolOp sibling overwrite their string per operand: on
x = [-2, -.5, .5, 1.5] `x > 0 and x < 1` counts 3 rows, `-1 < x <= 1` counts 3), and `not`, which vaex refuses (expresso.py validate:
"Unary op not supported").  A drop-in must not "fix" those: found by tests/test_vaex_differential.py.

and returns the columns, the comparison terms and the 16-bit truth table over the terms' outcomes that the C-ABI takes.
Anything else raises Unsupported — the caller then evaluates the mask on the host exactly as before."""
import ast
import itertools
import operator

import numpy as np

MAX_TERMS = 4
MAX_COLUMNS = 4
_DTYPES = ("float64", "float32", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool")
_OPS = {ast.Lt: 0, ast.LtE: 1, ast.Gt: 2, ast.GtE: 3, ast.Eq: 4, ast.NotEq: 5}
_SWAP = {0: 2, 1: 3, 2: 0, 3: 1, 4: 4, 5: 5}  # c < x  ==  x > c
_NUMPY = {0: operator.lt, 1: operator.le, 2: operator.gt, 3: operator.ge, 4: operator.eq, 5: operator.ne}


class Unsupported(ValueError):
    pass


def dtype_code(dtype):
    name = str(dtype).replace("torch.", "")
    if name not in _DTYPES:
        raise Unsupported(f"dtype {name} is not compared on the device")
    return _DTYPES.index(name)


def plain_numeric_dtype(ar):
    """numpy dtype of a column OBJECT when it is a real numeric column without missing values — a numpy array (1-d, native byte order,
    not masked) or a pyarrow Array / ChunkedArray of a primitive numeric / bool type with null_count == 0 (what vaex holds for
    arrow / parquet files: vaex/arrow/dataset.py) — else None.  Such a column's chunks reach the task part as arrays numpy views
    without a copy (bool: unpacked), so a comparison over it can run as a device predicate."""
    if isinstance(ar, np.ndarray):
        if np.ma.isMaskedArray(ar) or ar.ndim != 1 or not ar.dtype.isnative or ar.dtype.name not in _DTYPES:
            return None
        return ar.dtype
    if getattr(ar, "type", None) is None or not hasattr(ar, "null_count"):
        return None
    try:
        import pyarrow as pa
    except ImportError:
        return None
    if not isinstance(ar, (pa.Array, pa.ChunkedArray)) or ar.null_count != 0:
        return None
    try:
        dt = np.dtype(ar.type.to_pandas_dtype())
    except (TypeError, NotImplementedError, ValueError):
        return None
    return dt if dt.name in _DTYPES else None


class Predicate:
    """columns: names; terms: (column index, op code, python int/float constant); truth: bit b = keep when term outcomes spell b"""

    def __init__(self, expression, columns, terms, truth):
        self.expression, self.columns, self.terms, self.truth = expression, columns, terms, truth

    def key(self):
        return (tuple(self.columns), tuple(self.terms), self.truth)

    def numpy_mask(self, arrays):
        """the same predicate evaluated with numpy on host columns: what vaex itself does per chunk; used by the passes that take
        a ready-made mask (minmax, hashed groupby) and by the tests as the expected row set"""
        outcomes = []
        for c, op, value in self.terms:
            with np.errstate(invalid="ignore"):
                outcomes.append(_NUMPY[op](np.asarray(arrays[self.columns[c]]), value))
        bits = np.zeros(len(outcomes[0]), dtype=np.uint32)
        for t, o in enumerate(outcomes):
            bits |= o.astype(np.uint32) << t
        return ((self.truth >> bits) & 1).astype(bool)


def _constant(node):
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
        return node.value
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _constant(node.operand)
        if v is not None:
            return -v if isinstance(node.op, ast.USub) else v
    return None


def compile_selection(expression, known_columns):
    """expression string -> Predicate; raises Unsupported for anything outside the subset"""
    try:
        tree = ast.parse(expression.strip(), mode="eval").body
    except SyntaxError as e:
        raise Unsupported(str(e))
    columns, terms = [], []

    def term(name, op, value):
        if name not in known_columns:
            raise Unsupported(f"{name!r} is not a column")
        if isinstance(value, int) and not -(1 << 63) <= value < (1 << 63):
            raise Unsupported("integer constant out of range")
        if name not in columns:
            columns.append(name)
        t = (columns.index(name), op, value)
        if t not in terms:
            terms.append(t)
        return ("term", terms.index(t))

    def walk(node):
        if isinstance(node, ast.BoolOp):
            raise Unsupported("`and` / `or`: vaex keeps only the last operand of those; write & / |")
        if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.BitAnd, ast.BitOr)):
            return ("and" if isinstance(node.op, ast.BitAnd) else "or", [walk(node.left), walk(node.right)])
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.Invert):
            return ("not", [walk(node.operand)])
        if isinstance(node, ast.Compare):
            if len(node.ops) > 1:
                # vaex means the LAST link only, Python both: neither guess is taken here (module docstring)
                raise Unsupported("chained comparison: write (a < x) & (x <= b)")
            parts = []
            left = node.left
            for op, right in zip(node.ops, node.comparators):
                if type(op) not in _OPS:
                    raise Unsupported("comparison not supported")
                code = _OPS[type(op)]
                if isinstance(left, ast.Name) and _constant(right) is not None:
                    parts.append(term(left.id, code, _constant(right)))
                elif isinstance(right, ast.Name) and _constant(left) is not None:
                    parts.append(term(right.id, _SWAP[code], _constant(left)))
                else:
                    raise Unsupported("only <column> <op> <number> comparisons run on the device")
                left = right
            return parts[0] if len(parts) == 1 else ("and", parts)
        raise Unsupported(f"{type(node).__name__} is not part of the device predicate subset")

    tree = walk(tree)
    if not terms or len(terms) > MAX_TERMS or len(columns) > MAX_COLUMNS:
        raise Unsupported("more than 4 comparisons or columns")

    def value(node, outcome):
        kind, arg = node
        if kind == "term":
            return outcome[arg]
        if kind == "not":
            return not value(arg[0], outcome)
        vals = [value(a, outcome) for a in arg]
        return all(vals) if kind == "and" else any(vals)

    truth = 0
    for outcome in itertools.product([False, True], repeat=len(terms)):
        bits = sum(1 << t for t, o in enumerate(outcome) if o)
        if value(tree, outcome):
            truth |= 1 << bits
    return Predicate(expression, columns, terms, truth)
