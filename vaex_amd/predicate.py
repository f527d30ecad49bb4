"""Selection expressions -> device-side predicates (include/vaex_hip.h "device-side selections").

vaex keeps a selection as a boolean expression string (vaex/selections.py:50-76 SelectionExpression.boolean_expression,
validated against the AST whitelist of vaex/expresso.py:46-155) and evaluates it with numpy on every chunk
(vaex/execution.py:530-549).  `compile_selection` accepts the subset the GPU evaluates itself:

    comparisons  <column> (< <= > >= == !=) <number>,  <number> (op) <column>
    round 6: <expression> (op) <expression> over float64 columns (`x > y`, `x + y <= 2 * z`): both sides and the comparison in one program
    combined with  &  |  ~,  at most 4 comparisons over at most 4 columns
    round 5: the column side may be an ARITHMETIC EXPRESSION over float64 columns — + - * / unary minus, `** 2`, sqrt(), abs(),
    numbers — e.g. `2*x + 1 > 0`, `x**2 + y**2 < 4`; names of VIRTUAL columns (`virtual=`: name -> expression string, as in
    df.virtual_columns) are inlined, so a selection over `r = sqrt(x**2 + y**2)` qualifies.  Only operations whose float64 results are
    correctly rounded are taken (what numpy computes for them is what the device computes, bit for bit); other dtypes and functions
    stay with vaex's numpy evaluation (numpy's promotion rules and libm are not restated).

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


# steps of an expression term's postfix program (include/vaex_hip.h vxh_sel_op)
SEL_COL, SEL_CONST, SEL_ADD, SEL_SUB, SEL_MUL, SEL_DIV, SEL_NEG, SEL_SQUARE, SEL_SQRT, SEL_ABS = range(10)
SEL_CMP = 10   # round 6: SEL_CMP + comparison code (0 .. 5: < <= > >= == !=) compares the two top entries -> 1.0 / 0.0
MAX_STEPS = 16


class Predicate:
    """columns: names; terms: (column index, op code, python int/float constant); truth: bit b = keep when term outcomes spell b;
    programs: {term index: ((step op, column index, float value), ...)} for the terms whose left side is an arithmetic expression
    (their `column index` in `terms` is the first column the program reads)"""

    def __init__(self, expression, columns, terms, truth, programs=None):
        self.expression, self.columns, self.terms, self.truth = expression, columns, terms, truth
        self.programs = dict(programs or {})

    def key(self):
        return (tuple(self.columns), tuple(self.terms), self.truth, tuple(sorted(self.programs.items())))

    def _evaluate_program(self, steps, arrays):
        stack = []
        with np.errstate(all="ignore"):
            for op, c, value in steps:
                if op == SEL_COL:
                    stack.append(np.asarray(arrays[self.columns[c]], dtype=np.float64))
                elif op == SEL_CONST:
                    stack.append(np.float64(value))
                elif op in (SEL_ADD, SEL_SUB, SEL_MUL, SEL_DIV):
                    b, a = stack.pop(), stack.pop()
                    stack.append({SEL_ADD: operator.add, SEL_SUB: operator.sub, SEL_MUL: operator.mul, SEL_DIV: operator.truediv}[op](a, b))
                elif op >= SEL_CMP:
                    b, a = stack.pop(), stack.pop()
                    stack.append(np.asarray(_NUMPY[op - SEL_CMP](a, b), dtype=np.float64))
                elif op == SEL_NEG:
                    stack.append(-stack.pop())
                elif op == SEL_SQUARE:
                    a = stack.pop()
                    stack.append(a * a)
                elif op == SEL_SQRT:
                    stack.append(np.sqrt(stack.pop()))
                else:
                    stack.append(np.abs(stack.pop()))
        return stack[0]

    def torch_mask(self, tensors):
        """the same predicate over DEVICE columns (torch tensors), one elementwise float64 operation per program step — each of them
        (+ - * / negate, square as a * a, sqrt, abs) is correctly rounded on the device as in numpy, and no two are fused, so the rows
        kept are numpy_mask's.  For the passes that take a ready-made keep-mask over device columns (minmax, percentiles, hashed groupby)."""
        import torch
        ops = {0: torch.lt, 1: torch.le, 2: torch.gt, 3: torch.ge, 4: torch.eq, 5: torch.ne}
        bits = None
        for t, (c, op, value) in enumerate(self.terms):
            if t in self.programs:
                stack = []
                for step, sc, sv in self.programs[t]:
                    if step == SEL_COL:
                        stack.append(tensors[self.columns[sc]].to(torch.float64))
                    elif step == SEL_CONST:
                        first = next(iter(tensors.values()))
                        stack.append(torch.tensor(float(sv), dtype=torch.float64, device=first.device))
                    elif step in (SEL_ADD, SEL_SUB, SEL_MUL, SEL_DIV):
                        b, a = stack.pop(), stack.pop()
                        stack.append({SEL_ADD: torch.add, SEL_SUB: torch.sub, SEL_MUL: torch.mul, SEL_DIV: torch.true_divide}[step](a, b))
                    elif step >= SEL_CMP:
                        b, a = stack.pop(), stack.pop()
                        stack.append(ops[step - SEL_CMP](a, b).to(torch.float64))
                    elif step == SEL_NEG:
                        stack.append(torch.neg(stack.pop()))
                    elif step == SEL_SQUARE:
                        a = stack.pop()
                        stack.append(torch.mul(a, a))
                    elif step == SEL_SQRT:
                        stack.append(torch.sqrt(stack.pop()))
                    else:
                        stack.append(torch.abs(stack.pop()))
                col = stack[0]
            else:
                col = tensors[self.columns[c]]
            if isinstance(value, float) and not col.dtype.is_floating_point:
                col = col.to(torch.float64)  # numpy compares an integer column with a float constant in float64 (torch would pick float32)
            o = ops[op](col, value).to(torch.int32) << t
            bits = o if bits is None else bits | o
        return ((self.truth >> bits) & 1).to(torch.uint8)

    def numpy_mask(self, arrays):
        """the same predicate evaluated with numpy on host columns: what vaex itself does per chunk; used by the passes that take
        a ready-made mask (minmax, hashed groupby) and by the tests as the expected row set"""
        outcomes = []
        for t, (c, op, value) in enumerate(self.terms):
            with np.errstate(invalid="ignore"):
                left = self._evaluate_program(self.programs[t], arrays) if t in self.programs else np.asarray(arrays[self.columns[c]])
                outcomes.append(_NUMPY[op](left, value))
        bits = np.zeros(len(outcomes[0]), dtype=np.uint32)
        for t, o in enumerate(outcomes):
            bits |= o.astype(np.uint32) << t
        return ((self.truth >> bits) & 1).astype(bool)


def _constant(node, depth=0):
    """the value of a subtree made of numbers only, folded the way vaex's eval() folds it — Python semantics: integers stay exact
    integers (`9007199254740992 + 1 + 1` is ...994, not the ...992 step-by-step float64 folding gives), `/` is true division — or None.
    Function calls are never folded (vaex's sqrt / abs are numpy's)."""
    if depth > 24:
        return None
    if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
        return node.value
    if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
        v = _constant(node.operand, depth + 1)
        if v is not None:
            return -v if isinstance(node.op, ast.USub) else v
    if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult, ast.Div, ast.Pow)):
        a = _constant(node.left, depth + 1)
        b = _constant(node.right, depth + 1) if a is not None else None
        if b is None:
            return None
        try:
            if isinstance(node.op, ast.Pow):
                if not -64 <= b <= 64 or isinstance(a, int) and abs(a) > 1 << 64:   # (an exact integer power can take forever)
                    return None
                v = a ** b
            else:
                v = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv}[type(node.op)](a, b)
        except (ZeroDivisionError, OverflowError, ValueError):
            return None   # (vaex's eval raises on these: not a selection the device takes)
        if isinstance(v, complex) or isinstance(v, int) and v.bit_length() > 4096:
            return None
        return v
    return None


def compile_selection(expression, known_columns, virtual=None):
    """expression string -> Predicate; raises Unsupported for anything outside the subset.
    known_columns: name -> column object (its dtype decides whether it may appear in an arithmetic expression: float64 only);
    virtual: name -> expression string of the frame's virtual columns (inlined)"""
    try:
        tree = ast.parse(expression.strip(), mode="eval").body
    except SyntaxError as e:
        raise Unsupported(str(e))
    columns, terms, programs = [], [], {}
    virtual = dict(virtual or {})

    def is_f64(name):
        col = known_columns[name] if hasattr(known_columns, "__getitem__") else None
        dt = getattr(col, "dtype", None)
        if dt is None:
            dt = plain_numeric_dtype(col)
        return str(dt).replace("torch.", "") == "float64"

    def arithmetic(node, steps, depth=0):
        """postfix steps of an arithmetic expression over float64 columns; returns the stack depth it needs"""
        if depth > 24:
            raise Unsupported("expression nested too deeply")
        c = _constant(node)
        if c is not None:
            try:
                steps.append((SEL_CONST, 0, float(c)))   # (ONE rounding of the exactly folded constant, like numpy's conversion of a Python int)
            except OverflowError:
                raise Unsupported("integer constant too large for float64")
            return 1
        if not any(isinstance(n, ast.Name) for n in ast.walk(node)):
            # numbers only, and _constant did not fold them: `1 / 0`, `10 ** 400 * 1.0`, ... — vaex's eval() raises on these, the device would
            # quietly compute inf
            raise Unsupported("a constant subexpression Python does not evaluate")
        if isinstance(node, ast.Name):
            if node.id in virtual and node.id not in known_columns:
                try:
                    sub = ast.parse(str(virtual[node.id]).strip(), mode="eval").body
                except SyntaxError as e:
                    raise Unsupported(str(e))
                return arithmetic(sub, steps, depth + 1)
            if node.id not in known_columns:
                raise Unsupported(f"{node.id!r} is not a column")
            if not is_f64(node.id):
                raise Unsupported(f"arithmetic over {node.id!r}: only float64 columns are computed on the device (numpy's promotion rules stay on the host)")
            if node.id not in columns:
                columns.append(node.id)
            steps.append((SEL_COL, columns.index(node.id), 0.0))
            return 1
        if isinstance(node, ast.BinOp):
            if isinstance(node.op, ast.Pow):
                if _constant(node.right) != 2 or isinstance(_constant(node.right), float) and _constant(node.right) != 2.0:
                    raise Unsupported("only `** 2` is computed on the device (numpy squares; other powers go through libm)")
                d = arithmetic(node.left, steps, depth + 1)
                steps.append((SEL_SQUARE, 0, 0.0))
                return d
            code = {ast.Add: SEL_ADD, ast.Sub: SEL_SUB, ast.Mult: SEL_MUL, ast.Div: SEL_DIV}.get(type(node.op))
            if code is None:
                raise Unsupported(f"operator {type(node.op).__name__} is not computed on the device")
            dl = arithmetic(node.left, steps, depth + 1)
            dr = arithmetic(node.right, steps, depth + 1)
            steps.append((code, 0, 0.0))
            return max(dl, 1 + dr)
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            d = arithmetic(node.operand, steps, depth + 1)
            if isinstance(node.op, ast.USub):
                steps.append((SEL_NEG, 0, 0.0))
            return d
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in ("sqrt", "abs") and len(node.args) == 1 and not node.keywords:
            d = arithmetic(node.args[0], steps, depth + 1)
            steps.append((SEL_SQRT if node.func.id == "sqrt" else SEL_ABS, 0, 0.0))
            return d
        raise Unsupported(f"{type(node).__name__} is not part of the device expression subset")

    def expression_term(node, op, value):
        steps = []
        need = arithmetic(node, steps)
        if need > 4 or len(steps) > MAX_STEPS:
            raise Unsupported("expression too long for the device (16 steps, four stack entries)")
        if len(steps) == 1 and steps[0][0] == SEL_COL:   # (a virtual column that IS a real column)
            return term(columns[steps[0][1]], op, value)
        first = next((c for o, c, _ in steps if o == SEL_COL), None)
        if first is None:
            raise Unsupported("a comparison between two constants")
        try:
            t = (first, op, float(value))
        except OverflowError:
            raise Unsupported("integer constant too large for float64")
        prog = tuple(steps)
        for i, old in enumerate(terms):
            if old == t and programs.get(i) == prog:
                return ("term", i)
        terms.append(t)
        programs[len(terms) - 1] = prog
        return ("term", len(terms) - 1)

    def comparison_term(left, op, right):
        steps = []
        dl = arithmetic(left, steps)
        dr = arithmetic(right, steps)
        steps.append((SEL_CMP + op, 0, 0.0))
        if max(dl, 1 + dr) > 4 or len(steps) > MAX_STEPS:
            raise Unsupported("expression too long for the device (16 steps, four stack entries)")
        first = next((c for o, c, _ in steps if o == SEL_COL), None)
        if first is None:
            raise Unsupported("a comparison between two constants")
        t, prog = (first, 5, 0.0), tuple(steps)
        for i, old in enumerate(terms):
            if old == t and programs.get(i) == prog:
                return ("term", i)
        terms.append(t)
        programs[len(terms) - 1] = prog
        return ("term", len(terms) - 1)

    def term(name, op, value):
        if name not in known_columns:
            raise Unsupported(f"{name!r} is not a column")
        if isinstance(value, int) and not -(1 << 63) <= value < (1 << 63):
            raise Unsupported("integer constant out of range")
        if name not in columns:
            columns.append(name)
        t = (columns.index(name), op, value)
        for i, old in enumerate(terms):
            if old == t and i not in programs:
                return ("term", i)
        terms.append(t)
        return ("term", len(terms) - 1)

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
                plain = lambda nd: isinstance(nd, ast.Name) and (nd.id in known_columns or nd.id not in virtual)
                if plain(left) and _constant(right) is not None:
                    parts.append(term(left.id, code, _constant(right)))
                elif plain(right) and _constant(left) is not None:
                    parts.append(term(right.id, _SWAP[code], _constant(left)))
                elif _constant(right) is not None and _constant(left) is None:
                    parts.append(expression_term(left, code, _constant(right)))
                elif _constant(left) is not None and _constant(right) is None:
                    parts.append(expression_term(right, _SWAP[code], _constant(left)))
                elif _constant(left) is None and _constant(right) is None:
                    # round 6: <expression> <op> <expression> over float64 columns (`x > y`, `x + y <= 2 * z`): both sides and the comparison
                    # in ONE program that leaves 1.0 / 0.0 — the term is `<program> != 0`
                    parts.append(comparison_term(left, code, right))
                else:
                    raise Unsupported("only <column or arithmetic expression> <op> <number or expression> comparisons run on the device")
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
    return Predicate(expression, columns, terms, truth, programs)
