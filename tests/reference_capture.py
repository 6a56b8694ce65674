"""What `GPUDataCapture::from_evaluator` records (src/gpu_synthesizer/mod.rs:210-352, 385-444), emulated for a few evaluators:
every arithmetic call of `GpuSynthesizerFieldLike` takes a FRESH temporary from a process-wide counter (so a capture's
temporaries start wherever the previous captures stopped and are never reused), relations come in the order `evaluate_once`
issues its calls, `mul_and_accumulate_into` is the trait's default (a product, then a sum: field_like.rs:71-75).

`to_program` is the Python mirror of `OwnedProgram::from_capture` in rust/prove_hip.rs (dense renumbering in definition order,
field constants into the value table): captures in the reference's own order and numbering must evaluate like the
golden-pinned formulas of oracle/gates.py, on the CPU and through the device interpreter."""
from era_boojum_amd import gate_program as GP

P = GP.P
_counter = [977]            # as if other gates had been captured before in this process


class V:
    def __init__(self, idx):
        self.idx = idx      # ("var" | "wit" | "const" | "tmp" | "value", number)


class Capture:
    def __init__(self):
        self.relations, self.writes = [], []

    def _fresh(self, rel):
        _counter[0] += 1
        t = V(("tmp", _counter[0]))
        self.relations.append((t.idx, rel))
        return t
    # PrimeFieldLike for GpuSynthesizerFieldLike: `x.op_assign(&y)` rebinds x to a fresh temporary
    def add(self, x, y): return self._fresh(("Add", x.idx, y.idx))
    def sub(self, x, y): return self._fresh(("Sub", x.idx, y.idx))
    def mul(self, x, y): return self._fresh(("Mul", x.idx, y.idx))
    def square(self, x): return self._fresh(("Square", x.idx))
    def negate(self, x): return self._fresh(("Negate", x.idx))
    def double(self, x): return self._fresh(("Double", x.idx))
    def inverse(self, x): return self._fresh(("Inverse", x.idx))
    def mul_and_accumulate_into(self, acc, a, b): return self.add(acc, self.mul(a, b))      # field_like.rs:71-75
    def var(self, i): return V(("var", i))
    def wit(self, i): return V(("wit", i))
    def const(self, i): return V(("const", i))
    def value(self, x): return V(("value", x % P))
    def push(self, x): self.writes.append(x.idx)


def capture_fma():          # fma_gate_without_constant.rs:96-126
    c = Capture()
    a, b, cc, d = (c.var(i) for i in range(4))
    q, l = c.const(0), c.const(1)
    contribution = c.mul(cc, l)
    t = c.mul(a, b)
    contribution = c.mul_and_accumulate_into(contribution, q, t)
    contribution = c.sub(contribution, d)
    c.push(contribution)
    return c


def capture_zero_check(use_witness_column_for_inversion=False):      # zero_check.rs:143-175
    c = Capture()
    one = c.value(1)
    inp, flag = c.var(0), c.var(1)
    inv = c.wit(0) if use_witness_column_for_inversion else c.var(2)
    contribution = c.mul_and_accumulate_into(flag, inp, inv)
    contribution = c.sub(contribution, one)
    c.push(contribution)
    c.push(c.mul(inp, flag))
    return c


def capture_uintx_add():    # uintx_add.rs:101-140
    c = Capture()
    shift = c.const(0)
    a, b, carry_in, cc, carry_out = (c.var(i) for i in range(5))
    contribution = c.add(a, b)
    contribution = c.add(contribution, carry_in)
    contribution = c.sub(contribution, cc)
    tmp = c.mul(shift, carry_out)
    contribution = c.sub(contribution, tmp)
    c.push(contribution)
    contribution = c.mul(carry_out, carry_out)
    contribution = c.sub(contribution, carry_out)
    c.push(contribution)
    return c


def to_program(cap):
    """rust/prove_hip.rs::OwnedProgram::from_capture in Python: temporaries renumbered densely in definition order."""
    OPS = {"Add": GP.OP_ADD, "Sub": GP.OP_SUB, "Mul": GP.OP_MUL, "Double": GP.OP_DOUBLE, "Negate": GP.OP_NEGATE,
           "Square": GP.OP_SQUARE, "Inverse": GP.OP_INVERSE}
    KIND = {"var": GP.IDX_VARIABLE, "wit": GP.IDX_WITNESS, "const": GP.IDX_CONSTANT_POLY}
    values, rename = [], {}

    def ix(idx, define=False):
        k, i = idx
        if k == "tmp":
            if define:
                rename.setdefault(i, len(rename))
            return (GP.IDX_TEMPORARY, rename[i])
        if k == "value":
            if i not in values:
                values.append(i)
            return (GP.IDX_VALUE, values.index(i))
        return (KIND[k], i)
    relations = []
    for dst, rel in cap.relations:
        a = ix(rel[1])
        b = ix(rel[2]) if len(rel) > 2 else (0, 0)
        relations.append((OPS[rel[0]], ix(dst, True)[1], a, b))
    writes = [ix(w) for w in cap.writes]
    return GP.GateProgram(relations, values, writes, len(rename))
