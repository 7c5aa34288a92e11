"""Site-program IR and encoder (see genjax_amd/csrc/gmx_program.h for the
binary format).  A `Graph` is filled by the tracer (tracer.py / static.py) in
program order; `compile_graph` removes dead nodes, allocates the <= 64
registers by live range and emits the uint32 blob `gmx_program_create` takes.

This is the build's counterpart of the reference's trace-time machinery
(`stage` -> ClosedJaxpr, src/genjax/_src/core/compiler/staging.py:286-298; the
stateful interpreter, interpreters/stateful.py:47-86): same job — turn one run
of the model's Python source into an ordered list of sample sites with their
argument expressions — different mechanism.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

MAGIC = 0x50584D47
VERSION = 2
MAX_REGS = 64          # operand codes below POOL_BASE; more than 31 needs the specialised kernel (no interpreter build)

F_GATHER, F_U8, F_BCAST, F_STEP, F_FLAT, F_IDX = 1, 2, 4, 8, 16, 32

# opcode numbers: keep in sync with gmx_program.h
OPC = dict(
    END=0, CONST=1, UNI=2, LDIN=3, LDTAB=4, STOUT=5, LDKEY=6, KDERIVE=7, KDERIVER=8, LDIDX=9, MOV=10,
    ADD=11, SUB=12, MUL=13, DIV=14, MIN=15, MAX=16, POW=17,
    NEG=20, ABS=21, EXP=22, LOG=23, LOG1P=24, SQRT=25, SIN=26, COS=27, TANH=28, SIGMOID=29,
    SOFTPLUS=30, FLOOR=31, LGAMMA=32, SQUARE=33, RECIP=34, CEIL=35, ROUND=36,
    FLT=40, FLE=41, FGT=42, FGE=43, FEQ=44, FNE=45,
    IEQ=46, INE=47, ILT=48, ILE=49, IGT=50, IGE=51,
    AND=52, OR=53, NOT=54, XOR=55, SEL=56, I2F=57, F2I=58,
    IADD=60, ISUB=61, IMUL=62, INEG=63,
    S_NORMAL=70, S_UNIFORM=71, S_FLIP=72, S_BERNL=73, S_BETA=74, S_CATSTEP=75, S_LOGGAMMA=76,
    L_NORMAL=80, L_UNIFORM=81, L_FLIP=82, L_BERNL=83, L_BETA=84,
    REDMAX=90, REDLSE=91,
    LOOP=100, ENDLOOP=101, LDT=102, KSPLITU=103,
)

UNARY = {"MOV", "NEG", "ABS", "EXP", "LOG", "LOG1P", "SQRT", "SIN", "COS", "TANH", "SIGMOID",
         "SOFTPLUS", "FLOOR", "CEIL", "ROUND", "LGAMMA", "SQUARE", "RECIP", "NOT", "I2F", "F2I", "INEG"}
BINARY = {"ADD", "SUB", "MUL", "DIV", "MIN", "MAX", "POW", "FLT", "FLE", "FGT", "FGE", "FEQ", "FNE",
          "IEQ", "INE", "ILT", "ILE", "IGT", "IGE", "AND", "OR", "XOR", "IADD", "ISUB", "IMUL"}
SAMPLER2 = {"S_NORMAL", "S_UNIFORM", "S_BETA"}      # args (key, a, b), imm = element counter
ELEM_INDEX = 0xFFFFFF      # a sampler immediate meaning "the particle's global index" (csrc/gmx_program.h: GMX_ELEM_INDEX)
ELEM_LOOP = 0xFFFFFE       # ... "the iteration number of the innermost counted loop" (GMX_ELEM_LOOP: a long vector-valued site)
SAMPLER1 = {"S_FLIP", "S_BERNL", "S_LOGGAMMA"}                    # args (key, a)
LOGPDF2 = {"L_NORMAL", "L_UNIFORM", "L_BETA"}       # args (x, a, b)
LOGPDF1 = {"L_FLIP", "L_BERNL"}                     # args (x, a)
EFFECT = {"STOUT", "REDMAX", "REDLSE", "LOOP", "ENDLOOP", "SETVAR"}
# LOOPVAR (args = (init,)): a value carried across the iterations of a counted loop — one register (two for a
# key) initialised before OP_LOOP, read inside the block, overwritten by SETVAR (args = (var, new value)) and
# readable after OP_ENDLOOP.  LDT: the iteration number.
# LDINX (args = (index,), slot, imm = element offset): OP_LDIN flagged GMX_F_IDX — element `index + imm` of a [T, n] leaf,
# the index a register: `means[z]` on a per-particle vector in memory, at any loop depth (engine.StepInput._read_at)
_NO_CSE = EFFECT | {"LDIN", "LDINX", "UNI", "CONST", "S_CATSTEP", "CATIDX", "LOOPVAR", "LDT", "COPY"}     # slots are unique; CONSTs have their own table
_FOLD = {"ADD", "SUB", "MUL", "DIV", "SQUARE", "NEG"}
# values recomputed at every use instead of being held in a register (see compile_graph)
REMAT_UNARY = {"LOG", "EXP", "NEG", "SQUARE", "SQRT", "RECIP", "I2F"}
REMAT_BINARY = {"ADD", "SUB", "MUL", "DIV"}


@dataclass(eq=False)
class Node:
    op: str
    args: tuple = ()
    imm: int = 0
    dtype: str = "f32"          # 'f32' | 'i32' | 'bool' | 'key' | 'cat' | 'none'
    flags: int = 0
    slot: int = 0
    idx: int = -1               # position in Graph.nodes

    @property
    def width(self):
        return 2 if self.dtype in ("key", "cat") else 1


@dataclass
class Graph:
    nodes: list = field(default_factory=list)
    n_in: int = 0
    n_out: int = 0
    n_uni: int = 0
    n_tab: int = 0
    _consts: dict = field(default_factory=dict)
    _cse: dict = field(default_factory=dict)

    def add(self, op, args=(), imm=0, dtype="f32", flags=0, slot=0) -> Node:
        """Append a node.  Pure nodes are hash-consed: tracing the same expression twice (e.g. the
        normaliser of a categorical scored once per category in an enumeration) yields ONE node."""
        args = tuple(args)
        if dtype == "f32" and op in _FOLD:
            folded = self._fold(op, args)
            if folded is not None:
                return folded
        key = None
        if op not in _NO_CSE:
            key = (op, tuple(a.idx if a is not None else -1 for a in args), int(imm) & 0xFFFFFFFF, dtype, flags, slot)
            hit = self._cse.get(key)
            if hit is not None:
                return hit
        n = Node(op, args, int(imm) & 0xFFFFFFFF, dtype, flags, slot, len(self.nodes))
        self.nodes.append(n)
        if key is not None:
            self._cse[key] = n
        return n

    def _fold(self, op, args):
        """Trace-time simplifications that cannot change a bit of the result: an operation on CONSTANTS whose IEEE result
        is correctly rounded on the host as on gfx950 (numpy float32 add / sub / mul / div, the same with
        -ffp-contract=off on both sides) is evaluated now; `x / 1`, `x * 1`, `x - (+0)` are x for every x (NaN, infinities
        and signed zeros included).  What this buys: a normal log-density written out in primitive operations
        (distributions._Normal.sym_logpdf) sheds the divisions by a unit scale and shares `loc / scale`, `log scale`
        between the elements of a vector-valued site."""
        def cf(n):
            return np.float32(struct.unpack("<f", struct.pack("<I", n.imm))[0]) if (n is not None and n.op == "CONST" and n.dtype == "f32") else None
        vals = [cf(a) for a in args]
        if all(v is not None for v in vals):
            with np.errstate(all="ignore"):
                if op == "ADD": r = vals[0] + vals[1]
                elif op == "SUB": r = vals[0] - vals[1]
                elif op == "MUL": r = vals[0] * vals[1]
                elif op == "DIV": r = vals[0] / vals[1]
                elif op == "SQUARE": r = vals[0] * vals[0]
                else: r = np.float32(-vals[0])           # NEG
            return self.const_bits(int(np.float32(r).view(np.uint32)), "f32")
        if op == "DIV" and vals[1] is not None and vals[1].view(np.uint32) == 0x3F800000:
            return args[0]
        if op == "MUL" and vals[1] is not None and vals[1].view(np.uint32) == 0x3F800000 and args[0].dtype == "f32":
            return args[0]
        if op == "MUL" and vals[0] is not None and vals[0].view(np.uint32) == 0x3F800000 and args[1].dtype == "f32":
            return args[1]
        if op == "SUB" and vals[1] is not None and vals[1].view(np.uint32) == 0 and args[0].dtype == "f32":
            return args[0]
        return None

    # leaves -----------------------------------------------------------------
    def const_bits(self, bits: int, dtype: str) -> Node:
        k = (bits & 0xFFFFFFFF, dtype)
        n = self._consts.get(k)
        if n is None:
            n = self.add("CONST", imm=bits, dtype=dtype)
            self._consts[k] = n
        return n

    def const_f32(self, x: float) -> Node:
        return self.const_bits(struct.unpack("<I", struct.pack("<f", float(np.float32(x))))[0], "f32")

    def const_i32(self, x: int, dtype="i32") -> Node:
        return self.const_bits(int(x) & 0xFFFFFFFF, dtype)

    def uniform(self, dtype: str) -> Node:
        n = self.add("UNI", imm=self.n_uni, dtype=dtype)
        self.n_uni += 1
        return n

    def input(self, dtype: str, flags: int = 0) -> Node:
        if dtype == "bool":
            flags |= F_U8
        n = self.add("LDIN", dtype=dtype, flags=flags, slot=self.n_in)
        self.n_in += 1
        return n

    def table(self) -> int:
        t = self.n_tab
        self.n_tab += 1
        return t

    def store(self, value: Node, step: bool = False) -> int:
        """STOUT of `value`; returns the output slot.  step: element t of a [T, n] leaf (inside a loop)."""
        flags = (F_U8 if value.dtype == "bool" else 0) | (F_STEP if step else 0)
        if step and len(self.loop_counts) >= 2:
            flags |= F_FLAT            # inside nested loops: element (t0, t1[, t2]) of a [T0, T1[, T2], n] leaf (row-major)
        slot = self.n_out
        self.n_out += 1
        self.add("STOUT", (value,), dtype="none", flags=flags, slot=slot)
        return slot

    # counted loop ------------------------------------------------------------
    @property
    def loop_counts(self):
        """trip counts of the counted loops being traced, outermost first (at most three: a long scan inside a large
        plate, a plate of plates of plates ...)"""
        return self.__dict__.setdefault("_loop_counts", [])

    @property
    def _in_loop(self):
        return bool(self.loop_counts)

    def loop_begin(self, count: int):
        if len(self.loop_counts) >= 3:
            raise NotImplementedError("counted loops nest three deep (a plate of plates of plates), not four")
        self.loop_counts.append(int(count))
        if len(self.loop_counts) >= 2:
            self.nested_loops = True
        self._cse.clear()              # a value computed before the loop is not "the same" as one recomputed inside
        self.loop_ids.append(self.add("LOOP", imm=int(count), dtype="none").idx)

    @property
    def loop_ids(self):
        """the LOOP nodes of the counted loops being traced, outermost first: WHICH loops are open, not only how long"""
        return self.__dict__.setdefault("_loop_ids", [])

    def loop_end(self):
        self.loop_counts.pop()
        self.loop_ids.pop()
        self.add("ENDLOOP", dtype="none")
        self._cse.clear()

    def loop_var(self, init: Node) -> Node:
        return self.add("LOOPVAR", (init,), dtype=init.dtype)

    def set_var(self, var: Node, value: Node):
        self.add("SETVAR", (var, value), dtype="none")
        # (what a loop-carried value is computed FROM: its LOOPVAR node names the initial value only —
        #  static._depends follows these so that `jnp.sum(xs)` over a changed `xs` is seen to change)
        self.__dict__.setdefault("_var_updates", {}).setdefault(var.idx, []).append(value)

    def set_vars(self, pairs):
        """The loop-carried update as a PARALLEL copy: every new value is what it was BEFORE any variable of the group
        is overwritten (`(a, b) <- (x_new, a)`, a swap, an AR(2) shift register ...).  A new value that is itself a
        loop variable is first copied into a temporary; computed values already live in registers of their own."""
        pairs = [(var, val) for var, val in pairs if val is not var]
        vars_ = {var.idx for var, _ in pairs}
        staged = []
        for var, val in pairs:
            if val.op == "LOOPVAR" and val.idx in vars_:
                val = self.add("COPY", (val,), dtype=val.dtype)
            staged.append((var, val))
        for var, val in staged:
            self.set_var(var, val)


class ProgramTooLarge(Exception):
    pass


def _remat_leaves(a, remat):
    """The register-resident values an operand depends on (through rematerialised nodes)."""
    if a.idx not in remat:
        return [a]
    out = []
    for x in a.args:
        if x is not None:
            out += _remat_leaves(x, remat)
    return out


POOL_BASE = 64
POOL_SIZE = 64


def compile_graph(g: Graph):
    """DCE + linear-scan register allocation + encoding.

    Returns (blob uint32[], const_pool [(pool index, bits)]).  Launch uniforms
    and constants live in the operand pool (gmx_run_args.uni: uniforms first,
    constants after) and are named directly by source operands, so they cost
    no instruction and no register; constants that do not fit the 64-entry
    pool fall back to OP_CONST."""
    nodes = g.nodes
    # ---- liveness from effect roots ----
    live = [False] * len(nodes)
    stack = [n for n in nodes if n.op in EFFECT]
    while stack:
        n = stack.pop()
        if live[n.idx]:
            continue
        live[n.idx] = True
        for a in n.args:
            if a is not None and not live[a.idx]:
                stack.append(a)
    # ---- operand pool ----
    pool_of = {}
    const_pool = []
    next_pool = g.n_uni
    for n in nodes:
        if not live[n.idx]:
            continue
        if n.op == "UNI":
            pool_of[n.idx] = n.imm
        elif n.op == "CONST" and next_pool < POOL_SIZE:
            pool_of[n.idx] = next_pool
            const_pool.append((next_pool, n.imm))
            next_pool += 1
    # ---- rematerialisable values.  Hash-consing shares a value between distant uses (64
    # log-probabilities read once per enumerated category; y/sigma re-used by every gradient of an
    # HMC trajectory), which would pin a register each.  Cheap ones are re-emitted at each use instead:
    #   always:            a table element at a constant index, and one cheap unary op of it;
    #   if long-lived:     a non-gathered input load, and one cheap unary / binary op whose operands are
    #                      such loads or pool entries.
    def _first_last():
        first, last = {}, {}
        k = 0
        for n in nodes:
            if not live[n.idx]:
                continue
            for a in n.args:
                if a is not None:
                    first.setdefault(a.idx, k)
                    last[a.idx] = k
            k += 1
        return first, last
    use_first, use_last = _first_last()
    LONG = 96
    # nodes created between LOOP and ENDLOOP
    in_loop_region = {}
    inside = 0
    for n in nodes:
        if n.op == "LOOP":
            inside += 1
        elif n.op == "ENDLOOP":
            inside -= 1
        elif inside:
            in_loop_region[n.idx] = True

    def long_lived(n):
        return n.idx in use_last and use_last[n.idx] - use_first[n.idx] > LONG

    def leafish(a):
        return a is None or a.idx in pool_of or (a.idx in remat and remat[a.idx] == 1) or \
            (a.op == "CONST")
    remat = {}
    for n in nodes:
        if not live[n.idx]:
            continue
        if in_loop_region.get(n.idx):
            continue                     # inside a counted loop nothing is rematerialised: the block is short
        if n.op == "LDTAB" and n.args[0].op == "CONST":
            remat[n.idx] = 1
        elif n.op == "LDIN" and not (n.flags & F_GATHER) and long_lived(n):
            remat[n.idx] = 1
        elif n.op in REMAT_UNARY and n.args[0].idx in remat and remat[n.args[0].idx] == 1 and \
                (n.args[0].op == "LDTAB" or long_lived(n)):
            remat[n.idx] = 2
        elif n.op in REMAT_BINARY and long_lived(n) and all(leafish(a) for a in n.args) and \
                any(a.idx in remat for a in n.args):
            remat[n.idx] = 2
    # leaf loads (inputs, launch-index, register-resident constants) are SUNK to their first use:
    # a trace with many input leaves (an edited plate: value + score per element) would otherwise
    # hold every one of them in a register from the top of the program
    sunk = {n.idx for n in nodes if live[n.idx] and n.idx not in pool_of and n.op in ("LDIN", "CONST", "UNI", "LDIDX", "RELOAD2")}
    if getattr(g, "nested_loops", False):
        # with two loop levels a step-indexed load means "element t of the loop it was TRACED in": it stays where it is
        # (sunk into the inner loop, an outer-level load would read the inner iteration's element)
        sunk -= {n.idx for n in nodes if n.op == "LDIN" and n.flags & F_STEP}
    order = []
    placed = set()
    for n in nodes:
        if not live[n.idx] or n.idx in pool_of or n.idx in remat or n.idx in sunk:
            continue
        for a in n.args:
            if a is None:
                continue
            for leaf in _remat_leaves(a, remat):
                if leaf.idx in sunk and leaf.idx not in placed:
                    placed.add(leaf.idx)
                    order.append(leaf)
        order.append(n)
    last_use = {}
    for pos, n in enumerate(order):
        for a in n.args:
            if a is not None:
                for leaf in _remat_leaves(a, remat):
                    last_use[leaf.idx] = pos
    # counted loops: whatever is defined BEFORE the block and read inside it (and every loop-carried register) must
    # survive until the block has run for the last time — its register may not be handed to a value of the block
    pos_of = {n.idx: pos for pos, n in enumerate(order)}
    open_loops = []
    for pos, n in enumerate(order):
        if n.op == "LOOP":
            open_loops.append(pos)
        elif n.op == "ENDLOOP" and open_loops:
            loop_lo, loop_hi = open_loops.pop(), pos      # (an inner loop first, then the loop around it)
            for q in range(loop_lo + 1, loop_hi):
                for a in order[q].args:
                    if a is None:
                        continue
                    for leaf in _remat_leaves(a, remat):
                        if pos_of.get(leaf.idx, -1) < loop_lo and last_use.get(leaf.idx, -1) < loop_hi:
                            last_use[leaf.idx] = loop_hi
    # ---- registers ----
    free = [True] * MAX_REGS
    reg = {}
    n_regs = 0
    words = []

    def alloc(width):
        nonlocal n_regs
        for r in range(0, MAX_REGS - width + 1):
            if all(free[r + k] for k in range(width)):
                for k in range(width):
                    free[r + k] = False
                n_regs = max(n_regs, r + width)
                return r
        raise ProgramTooLarge(f"site program needs more than {MAX_REGS} live registers")

    def release(n):
        r = reg[n.idx]
        for k in range(n.width):
            free[r + k] = True

    def emit(op, dst=0, a=0, b=0, imm=0):
        words.append(OPC[op] | (dst & 0xFF) << 8 | (a & 0xFF) << 16 | (b & 0xFF) << 24)
        words.append(imm & 0xFFFFFFFF)

    temps = []
    pre = {}

    def scratch():
        t = alloc(1)
        temps.append(t)
        return t

    def operand(x):
        """Operand code of an argument of a value being rematerialised."""
        p = pool_of.get(x.idx)
        if p is not None:
            return POOL_BASE + p
        if x.idx in remat:
            return materialise(x)
        return reg[x.idx]

    def materialise(x):
        """Recompute a rematerialised value into a scratch register (freed after this instruction)."""
        if x.op == "LDTAB":
            t = scratch()
            emit("LDTAB", t, x.slot, operand(x.args[0]), x.imm)
        elif x.op == "LDIN":
            t = scratch()
            emit("LDIN", t, x.slot, x.flags, x.imm)
        elif x.op in UNARY:
            a = operand(x.args[0])
            t = a if a in temps else scratch()         # a scratch operand is updated in place
            emit(x.op, t, a)
        else:
            a, b = operand(x.args[0]), operand(x.args[1])
            t = a if a in temps else (b if b in temps else scratch())
            emit(x.op, t, a, b)
        return t

    def R(x):
        p = pool_of.get(x.idx)
        if p is not None:
            return POOL_BASE + p
        if x.idx in remat:
            return pre[x.idx]
        return reg[x.idx]

    def drop_temps():
        for t in temps:
            free[t] = True
        temps.clear()

    for pos, n in enumerate(order):
        drop_temps()
        # scratch registers for rematerialised operands are taken while every operand register is
        # still allocated, so they cannot alias a value this instruction reads
        pre.clear()
        for x in n.args:
            if x is not None and x.idx in remat and x.idx not in pre:
                pre[x.idx] = materialise(x)
        # operands whose last use is here may donate their registers to dst
        dying = [a for a in dict.fromkeys(l for x in n.args if x is not None for l in _remat_leaves(x, remat))
                 if last_use.get(a.idx) == pos and a.idx not in pool_of]
        op = n.op
        if op == "S_CATSTEP":
            prev, key, logit, ctr = n.args
            if prev is not None and prev in dying:
                dst = R(prev)
                dying.remove(prev)
            else:
                dst = alloc(2)
                if prev is not None:
                    emit("MOV", dst, R(prev)); emit("MOV", dst + 1, R(prev) + 1)
            emit(op, dst, R(logit), R(ctr), (R(key) & 0xFF) | (n.imm << 8))
            reg[n.idx] = dst
            for a in dying:
                release(a)
            continue
        for a in dying:
            release(a)
        if op in EFFECT:
            if op == "STOUT":
                # imm & 0xff: which word of a two-register value (split_graph); imm >> 8: element offset of a step store
                emit(op, n.flags, n.slot, R(n.args[0]) + (n.imm & 0xFF), n.imm >> 8)
            elif op == "LOOP":
                emit(op, imm=n.imm)
            elif op == "ENDLOOP":
                emit(op)
            elif op == "SETVAR":
                var, val = n.args
                if R(var) != R(val):
                    for k in range(var.width):
                        emit("MOV", R(var) + k, R(val) + k)
            else:
                emit(op, 0, R(n.args[0]))
            continue
        dst = alloc(n.width)
        reg[n.idx] = dst
        if op == "CONST":
            emit(op, dst, imm=n.imm)
        elif op == "UNI":
            emit(op, dst, imm=n.imm)
        elif op == "LDIN":
            emit(op, dst, n.slot, n.flags, n.imm)
        elif op == "LDTAB":
            emit(op, dst, n.slot, R(n.args[0]), n.imm)
        elif op == "LDINX":
            emit("LDIN", dst, n.slot, n.flags | F_IDX, (R(n.args[0]) & 0xFF) | ((n.imm & 0xFFFFFF) << 8))
        elif op in ("LDKEY", "LDIDX", "LDT"):
            emit(op, dst)
        elif op in ("LOOPVAR", "COPY"):
            init = n.args[0]
            for k in range(n.width):
                emit("MOV", dst + k, R(init) + k)
        elif op == "KDERIVE":
            emit(op, dst, R(n.args[0]), 0, n.imm)
        elif op in ("KDERIVER", "KSPLITU"):
            emit(op, dst, R(n.args[0]), R(n.args[1]))
        elif op == "RELOAD2":       # a two-register value (key, categorical state) read back from its two scratch words
            emit("LDIN", dst, n.slot, 0, 0)             # (split_graph: slot = the first word's input slot, imm = the second's)
            emit("LDIN", dst + 1, n.imm, 0, 0)
        elif op == "CATIDX":        # second register of a categorical state pair
            emit("MOV", dst, R(n.args[0]) + 1)
        elif op in UNARY:
            emit(op, dst, R(n.args[0]))
        elif op in BINARY:
            emit(op, dst, R(n.args[0]), R(n.args[1]))
        elif op == "SEL":
            c, a, b = n.args
            emit(op, dst, R(a), R(b), R(c))
        elif op in SAMPLER2:
            k, a, b = n.args
            emit(op, dst, R(a), R(b), (R(k) & 0xFF) | (n.imm << 8))
        elif op in SAMPLER1:
            k, a = n.args
            emit(op, dst, R(a), 0, (R(k) & 0xFF) | (n.imm << 8))
        elif op in LOGPDF2:
            x, a, b = n.args
            emit(op, dst, R(a), R(b), R(x))
        elif op in LOGPDF1:
            x, a = n.args
            emit(op, dst, R(a), 0, R(x))
        else:
            raise ValueError(f"unknown op {op}")
        if n.idx not in last_use:      # value never read (only possible for roots' helpers)
            release(n)
    n_instr = len(words) // 2
    consts = [bits & 0xFFFFFFFF for _, bits in const_pool]       # pool entries n_uni .. in order
    header = [MAGIC, VERSION, n_instr, max(n_regs, 1), g.n_in, g.n_out, next_pool, g.n_tab,
              len(consts), g.n_uni]
    return np.array(header + words + consts, dtype=np.uint32), const_pool


# ---------------------------------------------------------------------------
# programs that do not fit ONE launch: cut into a chain of launches
# ---------------------------------------------------------------------------
# The reference's handlers walk any number of `trace` sites (static.py:254-380) and XLA gives the fused computation
# whatever registers and buffers it needs.  A site program is bounded by the launch ABI (include/genmi.h: 64 input
# leaves, 64 output leaves, 64 pool entries, 8 tables) and by 64 live 32-bit values per particle.  A graph beyond those
# bounds is CUT, at top level (never inside a counted loop), into segments that each fit; a value computed in one
# segment and read in a later one is stored to a scratch leaf by the first ("spill") and loaded by the second — same
# nodes, same order, same arithmetic, so the chain computes bit for bit what the single program would.  Values that
# cost nothing to obtain again (constants, launch uniforms, input leaves, the particle's key and index) are re-issued
# in whichever segment needs them instead of being spilled.
FREE_OPS = ("CONST", "UNI", "LDIN", "LDKEY", "LDIDX")


@dataclass
class Segment:
    graph: Graph
    in_src: list          # local input slot  -> ("in", slot of the whole graph) | ("spill", scratch word)
    out_dst: list         # local output slot -> ("out", slot of the whole graph) | ("spill", scratch word)
    uni_src: list         # local launch uniform -> launch uniform of the whole graph
    tab_src: list         # local table -> table of the whole graph
    blob: np.ndarray = None
    const_pool: list = None
    has_red: bool = False


def split_graph(g: Graph, max_in: int, max_out: int, max_uni: int, max_tab: int, max_len: int = 2048):
    """Cut `g` into a chain of Segments (each compiled: .blob / .const_pool); returns (segments, n_spill_words).
    Raises ProgramTooLarge when ONE indivisible unit (a counted loop with the values it carries) does not fit."""
    nodes = g.nodes
    live = [False] * len(nodes)
    stack = [n for n in nodes if n.op in EFFECT]
    while stack:
        n = stack.pop()
        if live[n.idx]:
            continue
        live[n.idx] = True
        for a in n.args:
            if a is not None and not live[a.idx]:
                stack.append(a)
    # positions: live nodes in program order; the free ones traced at top level are re-issued on demand instead
    depth, P, free = 0, [], set()
    for n in nodes:
        if n.op == "ENDLOOP":
            depth -= 1
        if live[n.idx]:
            if n.op in FREE_OPS and depth == 0:
                free.add(n.idx)
            else:
                P.append(n)
        if n.op == "LOOP":
            depth += 1
    pos = {n.idx: p for p, n in enumerate(P)}
    last_use = {}
    for p, n in enumerate(P):
        for a in n.args:
            if a is not None and a.idx in pos:
                last_use[a.idx] = p
    # indivisible units: a top-level counted loop together with the loop-carried values defined in front of it
    units, p, depth = [], 0, 0
    while p < len(P):
        if P[p].op != "LOOP":
            units.append([p, p])
            p += 1
            continue
        s, depth, q = p, 0, p
        while True:
            if P[q].op == "LOOP":
                depth += 1
            elif P[q].op == "ENDLOOP":
                depth -= 1
                if depth == 0:
                    break
            q += 1
        for r in range(p + 1, q):
            if P[r].op == "SETVAR":
                v = P[r].args[0]
                if v.idx in pos and pos[v.idx] < s:
                    s = pos[v.idx]
        while units and units[-1][1] >= s:           # the carried values' definitions join the loop's unit
            s = min(s, units.pop()[0])
        units.append([s, q])
        p = q + 1

    spill_word = {}          # node idx -> first scratch word
    n_words = 0
    segments = []

    def build(ulo, uhi):
        """the sub-graph of units ulo .. uhi (inclusive), with its spills"""
        nonlocal n_words
        lo, hi = units[ulo][0], units[uhi][1]
        sg = Graph()
        sg.tables = []
        if getattr(g, "nested_loops", False):
            sg.nested_loops = True
        seg = Segment(sg, [], [], [], [])
        in_local, uni_local, tab_local, m = {}, {}, {}, {}
        gtabs = g.__dict__.get("tables", [])

        def put(op, args=(), imm=0, dtype="f32", flags=0, slot=0):
            nd = Node(op, tuple(args), int(imm) & 0xFFFFFFFF, dtype, flags, slot, len(sg.nodes))
            sg.nodes.append(nd)
            return nd

        def local_in(src):
            s_ = in_local.get(src)
            if s_ is None:
                s_ = in_local[src] = len(seg.in_src)
                seg.in_src.append(src)
            return s_

        def local_uni(u):
            s_ = uni_local.get(u)
            if s_ is None:
                s_ = uni_local[u] = len(seg.uni_src)
                seg.uni_src.append(u)
            return s_

        def local_tab(t):
            s_ = tab_local.get(t)
            if s_ is None:
                s_ = tab_local[t] = len(seg.tab_src)
                seg.tab_src.append(t)
                sg.tables.append(gtabs[t] if t < len(gtabs) else None)
            return s_

        def clone(n, args):
            if n.op == "UNI":
                return put("UNI", (), local_uni(n.imm), n.dtype)
            if n.op == "LDIN":
                return put("LDIN", (), n.imm, n.dtype, n.flags, local_in(("in", n.slot)))
            if n.op == "LDTAB":
                return put("LDTAB", args, n.imm, n.dtype, n.flags, local_tab(n.slot))
            if n.op == "LDINX":
                return put("LDINX", args, n.imm, n.dtype, n.flags, local_in(("in", n.slot)))
            if n.op == "STOUT":
                seg.out_dst.append(("out", n.slot))
                return put("STOUT", args, n.imm & ~0xFF, "none", n.flags, len(seg.out_dst) - 1)
            return put(n.op, args, n.imm, n.dtype, n.flags, n.slot)

        def operand(a):
            if a is None:
                return None
            nd = m.get(a.idx)
            if nd is not None:
                return nd
            if a.idx in free:
                nd = clone(a, ())
            else:                                   # computed by an earlier segment: read its scratch words
                w0 = spill_word[a.idx]
                if a.width == 1:
                    nd = put("LDIN", (), 0, a.dtype, 0, local_in(("spill", w0)))
                else:
                    s0, s1 = local_in(("spill", w0)), local_in(("spill", w0 + 1))
                    nd = put("RELOAD2", (), s1, a.dtype, 0, s0)
            m[a.idx] = nd
            return nd

        for u in range(ulo, uhi + 1):
            a_, b_ = units[u]
            for n in P[a_:b_ + 1]:                   # what the unit reads from outside itself: issued in front of it
                for x in n.args:
                    if x is not None and x.idx not in m and not (a_ <= pos.get(x.idx, -1) <= b_):
                        operand(x)
            for n in P[a_:b_ + 1]:
                m[n.idx] = clone(n, [operand(x) for x in n.args])
                if n.op in ("REDMAX", "REDLSE"):
                    seg.has_red = True
        for n in P[lo:hi + 1]:
            if last_use.get(n.idx, -1) > hi and n.dtype != "none":
                spill_word[n.idx] = n_words
                for k in range(n.width):
                    seg.out_dst.append(("spill", n_words + k))
                    put("STOUT", (m[n.idx],), k, "none", 0, len(seg.out_dst) - 1)
                n_words += n.width
        sg.n_in, sg.n_out, sg.n_uni, sg.n_tab = len(seg.in_src), len(seg.out_dst), len(seg.uni_src), len(seg.tab_src)
        return seg

    ulo = 0
    while ulo < len(units):
        seg_lo = units[ulo][0]
        ins, unis, tabs, reloads = set(), set(), set(), set()
        stouts = cur = e_prev = 0
        e_prev = seg_lo - 1
        expire = {}
        best = None
        fits = []                # the cut points after which everything in flight fits the stored leaves
        for u in range(ulo, len(units)):
            a_, b_ = units[u]
            for n in P[a_:b_ + 1]:
                if n.op == "STOUT":
                    stouts += 1
                elif n.op == "LDTAB":
                    tabs.add(n.slot)
                elif n.op in ("LDIN", "LDINX"):
                    ins.add(("in", n.slot))
                elif n.op == "UNI":
                    unis.add(n.imm)
                for x in n.args:
                    if x is None:
                        continue
                    if x.idx in free:
                        if x.op == "LDIN":
                            ins.add(("in", x.slot))
                        elif x.op == "UNI":
                            unis.add(x.imm)
                    elif pos[x.idx] < seg_lo and x.idx not in reloads:
                        reloads.add(x.idx)
                        for k in range(x.width):
                            ins.add(("spill", x.idx, k))
                lu = last_use.get(n.idx, -1)
                if lu > b_ and n.dtype != "none":
                    cur += n.width
                    expire[lu] = expire.get(lu, 0) + n.width
            for q in range(e_prev + 1, b_ + 1):
                cur -= expire.pop(q, 0)
            e_prev = b_
            if len(ins) > max_in or len(unis) > max_uni or len(tabs) > max_tab or stouts > max_out:
                break
            if stouts + cur <= max_out:
                best = u
                fits.append(u)
            if b_ - seg_lo >= max_len and best is not None:
                break
        if best is None:
            raise ProgramTooLarge("one indivisible part of the site program (a counted loop with the values it carries, "
                                  "or a single site) exceeds the launch slots")
        while True:
            mark = (n_words, dict(spill_word))
            seg = build(ulo, best)
            try:
                seg.blob, seg.const_pool = compile_graph(seg.graph)
                break
            except ProgramTooLarge:
                n_words = mark[0]
                spill_word.clear()
                spill_word.update(mark[1])
                if best == ulo:
                    raise
                # a shorter segment — cut only where the values in flight fit the stored leaves (a cut half way may
                # fall where more of them are alive than at the end that was tried)
                shorter = [u for u in fits if u <= ulo + (best - ulo) // 2]
                if not shorter:
                    shorter = [u for u in fits if u < best]
                if not shorter:
                    raise
                best = shorter[-1]
        segments.append(seg)
        ulo = best + 1
    return segments, n_words
