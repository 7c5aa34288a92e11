"""Reverse-mode differentiation of the site-program IR.

`grad(out, wrt)` returns d out / d w for each Expr w in `wrt` as new expressions in the same
graph (the tracer's operators build them), so a gradient costs no extra launch: it is more
straight-line code in the program that needed it.  This is what `jax.grad(assess)` is to the
reference's HMC (src/genjax/_src/inference/requests/hmc.py:69-97).

The `wrt` nodes are treated as independent inputs: propagation stops at them.  Integer / boolean
ops, comparisons and table indices carry no gradient.  Density ops differentiate in closed form:
  L_NORMAL(x; m, s):  z = x/s - m/s;  d/dx = -z/s,  d/dm = z/s,  d/ds = (z*z - 1)/s
(the derivative of TFP's  -0.5*squared_difference(x/s, m/s) - log s - 0.5*log(2 pi)).
"""
from __future__ import annotations

from . import tracer as T
from .tracer import Expr
from .engine import new_cache as _new_program_cache


class NotDifferentiable(NotImplementedError):
    pass


def grad(out: Expr, wrt: list) -> list:
    g = T.current_graph()
    wrt_idx = {w.node.idx for w in wrt}
    # a loop-carried SUM whose loop accumulated its own derivatives (static._vector_site_loop: the score of a long
    # vector-valued site, d / d w summed over the elements in the same loop): var idx -> [(w node, derivative node)]
    custom = g.__dict__.get("_custom_grads", {})
    # nodes that depend on a wrt node (forward reachability), restricted to ancestors of `out`
    anc = set()
    stack = [out.node]
    while stack:
        n = stack.pop()
        if n.idx in anc:
            continue
        anc.add(n.idx)
        if n.idx in wrt_idx:
            continue
        stack.extend(a for a in n.args if a is not None)
        if n.op == "LOOPVAR" and n.idx in custom:
            stack.extend(wn for wn, _ in custom[n.idx])
    dep = set(wrt_idx)
    order = sorted(anc)
    by_idx = {i: g.nodes[i] for i in order}
    for i in order:                              # args precede their users in the node list
        n = by_idx[i]
        if i not in dep and any(a is not None and a.idx in dep for a in n.args):
            dep.add(i)
        if i not in dep and n.op == "LOOPVAR" and any(wn.idx in dep for wn, _ in custom.get(i, ())):
            dep.add(i)
    adj = {out.node.idx: T.lift(1.0)}

    def push(node, contribution):
        if node is None or node.idx not in dep:
            return
        cur = adj.get(node.idx)
        adj[node.idx] = contribution if cur is None else cur + contribution

    for i in reversed(order):
        if i not in dep or i in wrt_idx:
            continue
        a_ = adj.get(i)
        if a_ is None:
            continue
        n = by_idx[i]
        if n.op == "LOOPVAR":
            push(n.args[0], a_)                    # the initial value
            for wn, gv in custom.get(i, ()):
                push(wn, a_ * Expr(gv))            # d (sum over the loop) / d w, accumulated by the loop itself
            continue
        _rule(n, a_, push)
    zero = T.lift(0.0)
    return [adj.get(w.node.idx, zero) for w in wrt]


def _rule(n, a, push):
    op = n.op
    E = lambda node: Expr(node)
    if op in ("MOV",):
        push(n.args[0], a)
    elif op == "ADD":
        push(n.args[0], a); push(n.args[1], a)
    elif op == "SUB":
        push(n.args[0], a); push(n.args[1], -a)
    elif op == "MUL":
        push(n.args[0], a * E(n.args[1])); push(n.args[1], a * E(n.args[0]))
    elif op == "DIV":
        x, y = E(n.args[0]), E(n.args[1])
        push(n.args[0], a / y); push(n.args[1], -(a * E(n)) / y)
    elif op == "NEG":
        push(n.args[0], -a)
    elif op == "ABS":
        x = E(n.args[0])
        push(n.args[0], T.where(x < 0.0, -a, a))
    elif op == "EXP":
        push(n.args[0], a * E(n))
    elif op == "LOG":
        push(n.args[0], a / E(n.args[0]))
    elif op == "LOG1P":
        push(n.args[0], a / (E(n.args[0]) + 1.0))
    elif op == "SQRT":
        push(n.args[0], a / (E(n) * 2.0))
    elif op == "SQUARE":
        push(n.args[0], a * (E(n.args[0]) * 2.0))
    elif op == "RECIP":
        push(n.args[0], -(a * E(n) * E(n)))
    elif op == "SIN":
        push(n.args[0], a * T.unary("COS")(E(n.args[0])))
    elif op == "COS":
        push(n.args[0], -(a * T.unary("SIN")(E(n.args[0]))))
    elif op == "TANH":
        push(n.args[0], a * (1.0 - E(n) * E(n)))
    elif op == "SIGMOID":
        push(n.args[0], a * (E(n) * (1.0 - E(n))))
    elif op == "SOFTPLUS":
        push(n.args[0], a * T.unary("SIGMOID")(E(n.args[0])))
    elif op == "POW":
        x, y = E(n.args[0]), E(n.args[1])
        push(n.args[0], a * (y * T.power(x, y - 1.0)))
        push(n.args[1], a * (E(n) * T.unary("LOG")(x)))
    elif op in ("MIN", "MAX"):
        x, y = E(n.args[0]), E(n.args[1])
        first = (x < y) if op == "MIN" else (x > y)
        push(n.args[0], T.where(first, a, 0.0)); push(n.args[1], T.where(first, 0.0, a))
    elif op == "SEL":
        c, x, y = n.args
        push(x, T.where(E(c), a, 0.0)); push(y, T.where(E(c), 0.0, a))
    elif op == "L_NORMAL":
        x, m, s = (E(v) for v in n.args)
        z = x / s - m / s
        push(n.args[0], -(a * (z / s)))
        push(n.args[1], a * (z / s))
        push(n.args[2], a * ((z * z - 1.0) / s))
    elif op == "L_BERNL":              # x*l - softplus(l):  d/dl = x - sigmoid(l)
        x, l = n.args
        push(l, a * (T.as_float(E(x)) - T.unary("SIGMOID")(E(l))))
    elif op == "L_FLIP":               # x log p + (1-x) log(1-p)
        x, p = n.args
        xf, pf = T.as_float(E(x)), E(p)
        push(p, a * (xf / pf - (1.0 - xf) / (1.0 - pf)))
    elif op == "L_UNIFORM":
        x, lo, hi = (E(v) for v in n.args)
        inside = T.where((x >= lo), 1.0, 0.0) * T.where((x <= hi), 1.0, 0.0)
        w = hi - lo
        push(n.args[1], a * (inside / w)); push(n.args[2], -(a * (inside / w)))
    elif op in ("FLOOR", "CEIL", "ROUND", "I2F", "F2I", "LDTAB", "LDIN", "UNI", "CONST", "LDIDX") or \
            op.startswith(("F", "I")) and op not in ("FLOOR",) or op in ("AND", "OR", "NOT", "XOR"):
        return                          # piecewise constant / not a float function of its inputs
    else:
        raise NotDifferentiable(f"no derivative rule for {op}")


# ---------------------------------------------------------------------------
# value_and_grad of a plain numeric function (the `jax.value_and_grad` of this stack): one launch
# ---------------------------------------------------------------------------
_VG_CACHE = _new_program_cache()


def value_and_grad(fn):
    """`value_and_grad(fn)(*xs)`: xs are float tensors of one common batch shape; returns
    (fn(*xs), (d fn / d x_k for each k)) computed per element in ONE launch."""
    def run(*xs):
        from . import _lib
        from .engine import Compiled, Flat, Tracing, leaf_spec, resolve
        from .static import _fnkey
        flat = Flat()
        tree = flat.add(tuple(xs))
        batch = tuple(xs[0].shape)
        specs = tuple(leaf_spec(v, batch) for v in flat.leaves)
        ck = (_fnkey(fn), tree, specs)
        ent = _VG_CACHE.get(ck)
        if ent is None:
            tr = Tracing(len(batch))
            with T.tracing(tr.graph):
                syms = [tr.sym_leaf(s, j) for j, s in enumerate(specs)]
                ins = [T.as_float(s.value) for s in syms]
                out = T.as_float(fn(*ins))
                gs = grad(out, ins)
                oo = tr.emit_output(out)
                go = [tr.emit_output(g + 0.0) for g in gs]
            ent = (Compiled(tr), oo, go)
            _VG_CACHE[ck] = ent
        comp, oo, go = ent
        outs = comp.run(flat.leaves, batch, None)
        _lib.get()
        return resolve(oo, outs, flat.leaves), tuple(resolve(g, outs, flat.leaves) for g in go)
    return run
