#!/usr/bin/env python3
"""Generates SUPPORT.md: which model FORM runs under which GFI METHOD at which SIZE, and how.

Every cell is RUN — on the CPU mirror of the C-ABI (tests/hostsim: the product's own interpreter template compiled for the
host; same tracer, same programs as on the GPU) — under a batch of 5 keys and under ONE key, and what happened is written
down: the executor the call took (programs launched, counted loops inside them, a chain of launches, the site-by-site
form) or the exception it raised.  Nothing in the table is prose from memory; `python tools/support_matrix.py` rewrites it,
and `tests/test_host_logic.py::test_support_matrix_is_current` fails when the committed file no longer matches a sample of
freshly run cells.

    python tools/support_matrix.py            # rewrites SUPPORT.md (about a minute)
    python tools/support_matrix.py --check    # exit 1 if SUPPORT.md differs from a fresh run
"""
from __future__ import annotations

import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

import numpy as np          # noqa: E402
import torch                # noqa: E402

METHODS = ["simulate", "importance", "assess", "update", "regenerate", "rejuvenate", "hmc", "index"]
B = 5


class Probe:
    """what a call launched: programs created / launched, loops, chains, the site-by-site form"""

    def __init__(self):
        from genjax_amd import engine, sitewise
        self.engine, self.sitewise = engine, sitewise
        self.launches = self.loops = self.chains = self.sitewise_calls = 0

    def __enter__(self):
        eng, sw, me = self.engine, self.sitewise, self
        self._launch, self._run_gfi = eng.Compiled.launch, sw.run_gfi
        self._run_edit = getattr(sw, "run_edit", None)

        def launch(comp, bound):
            me.launches += len(comp.links) if comp.links else 1
            me.chains += 1 if comp.links else 0
            blobs = [l.seg.blob for l in comp.links] if comp.links else [comp.blob]
            for bl in blobs:
                n_instr = int(bl[2])
                words = np.asarray(bl[10:10 + 2 * n_instr:2])
                me.loops += int(np.sum((words & 0xFF) == 100))
            return me._launch(comp, bound)

        def run_gfi(*a, **k):
            me.sitewise_calls += 1
            return me._run_gfi(*a, **k)
        eng.Compiled.launch = launch
        sw.run_gfi = run_gfi
        if self._run_edit is not None:
            def run_edit(*a, **k):
                me.sitewise_calls += 1
                return me._run_edit(*a, **k)
            sw.run_edit = run_edit
        return self

    def __exit__(self, *exc):
        self.engine.Compiled.launch = self._launch
        self.sitewise.run_gfi = self._run_gfi
        if self._run_edit is not None:
            self.sitewise.run_edit = self._run_edit

    def how(self):
        if self.sitewise_calls:
            return f"site by site ({self.launches} launches)"
        parts = [f"{self.launches} launch" + ("es" if self.launches != 1 else "")]
        if self.chains:
            parts.append("chain")
        if self.loops:
            parts.append(f"{self.loops} loop" + ("s" if self.loops != 1 else ""))
        return ", ".join(parts)


# ---------------------------------------------------------------------------------------------------------------------
# forms: name -> builder(size) -> dict(model, args, obs (constraint for importance), upd (constraint for update),
#                                      sel (address to regenerate / rejuvenate / HMC), index=(plate addr, site, size) | None)
# ---------------------------------------------------------------------------------------------------------------------
def forms():
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, numpy as jnp

    def scalar(_n):
        @G.gen
        def m(a):
            x = G.normal(a, 1.0) @ "x"
            y = G.normal(x * 0.5, 2.0) @ "y"
            return y
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["y"].set(1.0), upd=C["x"].set(0.3), sel="x")

    def vector_site(n):
        xs = np.linspace(-1, 1, n).astype(np.float32)

        @G.gen
        def m(a):
            s = G.normal(a, 1.0) @ "s"
            G.normal(s * jnp.array(xs), 0.5) @ "y"
            return s
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["y"].set(jnp.array(xs)), upd=C["s"].set(0.3), sel="s")

    def latent_vector(n):
        sig = np.linspace(9, 18, n).astype(np.float32)

        @G.gen
        def m(a):
            mu = G.normal(a, 5.0) @ "mu"
            th = G.normal(mu * jnp.ones(n), 2.0 * jnp.ones(n)) @ "theta"
            G.normal(th, jnp.array(sig)) @ "y"
            return mu
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["y"].set(jnp.array(sig)), upd=C["mu"].set(0.3), sel="theta")

    def gathered(n):
        """random effects: n group effects, 5 n observations each reading its group's effect"""
        N_ = 5 * n
        grp = (np.arange(N_) * 7 % n).astype(np.int32)

        @G.gen
        def m(a):
            mu = G.normal(a, 5.0) @ "mu"
            th = G.normal(mu * jnp.ones(n), 2.0 * jnp.ones(n)) @ "theta"
            G.normal(th[jnp.array(grp)], 0.5) @ "y"
            return mu
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["y"].set(jnp.array(np.zeros(N_, np.float32))), upd=C["mu"].set(0.3),
                    sel="theta")

    def plate(n):
        xs = np.linspace(-1, 1, n).astype(np.float32)

        @G.gen
        def elem(s, x):
            v = G.normal(s + x, 1.0) @ "v"
            return v

        @G.gen
        def m(a):
            s = G.normal(a, 1.0) @ "s"
            vs = elem.vmap(in_axes=(None, 0))(s, jnp.array(xs)) @ "p"
            return s
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["p", :, "v"].set(jnp.array(xs)), upd=C["s"].set(0.3), sel="s",
                    index=("p", "v", n))

    def bare_plate_used(n):
        @G.gen
        def m(a):
            x = G.normal(a, 1.0) @ "x"
            vs = G.normal.vmap()(jnp.zeros(n), jnp.ones(n)) @ "p"
            G.normal(jnp.sum(vs) + x + vs[1], 5.0) @ "obs"
            return x
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["obs"].set(1.0), upd=C["x"].set(0.3), sel="x")

    def plate_traced_index(n):
        @G.gen
        def elem(s):
            return G.normal(s, 1.0) @ "mean"

        @G.gen
        def m(a):
            means = elem.repeat(n=n)(a) @ "clusters"
            z = G.categorical(logits=jnp.zeros(3)) @ "z"
            G.normal(means[z], 1.0) @ "y"
            return z
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["y"].set(1.0), upd=C["z"].set(1), sel="z", hmc=False)

    def scan(T_):
        xs = np.linspace(-1, 1, T_).astype(np.float32)

        @G.gen
        def step(c, x):
            z = G.normal(c * 0.5 + x, 1.0) @ "z"
            return z, z

        @G.gen
        def m(a):
            cT, zs = step.scan(n=T_)(a, jnp.array(xs)) @ "s"
            G.normal(jnp.sum(zs) + cT, 3.0) @ "y"
            return cT
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["y"].set(1.0), upd=C["s", :, "z"].set(jnp.array(xs)), sel=("s", "z"),
                    index=("s", "z", T_))

    def scan_array_carry(T_):
        @G.gen
        def step(c, _):
            z = G.normal(c[0] * 0.5 + c[1] * 0.25, 1.0) @ "z"
            return jnp.stack([z, c[0]]), z

        @G.gen
        def m(a):
            cT, zs = step.scan(n=T_)(jnp.zeros(2), None) @ "s"
            G.normal(cT[0] + a, 3.0) @ "y"
            return zs
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["y"].set(1.0), upd=C["y"].set(0.2), sel=("s", "z"))

    def plate_of_scans(n):
        T_ = 20

        @G.gen
        def step(c, x):
            z = G.normal(c * 0.5 + x, 1.0) @ "z"
            return z, z

        @G.gen
        def elem(s, x):
            cT, _ = step.scan(n=T_)(s + x, jnp.array(np.linspace(-1, 1, T_).astype(np.float32))) @ "chain"
            return cT

        @G.gen
        def m(a):
            s = G.normal(a, 1.0) @ "s"
            elem.vmap(in_axes=(None, 0))(s, jnp.array(np.linspace(0, 1, n).astype(np.float32))) @ "p"
            return s
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["s"].set(0.1), upd=C["s"].set(0.3), sel="s", index=("p", ("chain",), n))

    def direct_plate_of_scans(n):
        """`kernel.scan(n=T).vmap()` as the model itself: its trace is the combinators' own (leaves [J, T])"""
        T_ = n

        @G.gen
        def step(c, x):
            z = G.normal(c * 0.5 + x, 1.0) @ "z"
            G.normal(z, 0.75) @ "y"
            return z, z
        J = 20
        m = step.scan(n=T_).vmap(in_axes=(0, None))
        xs = jnp.array(np.linspace(-1, 1, T_).astype(np.float32))
        return dict(model=m, args=lambda k: (jnp.array(np.linspace(0, 1, J).astype(np.float32)), xs),
                    obs=C["y"].set(jnp.array(np.zeros((J, T_), np.float32))), upd=C[1, 2, "z"].set(0.3), sel="z", hmc=False,
                    index=(None, "y", J))

    def masked(n):
        @G.gen
        def inner(s):
            return G.normal(s, 1.0) @ "y"

        @G.gen
        def m(a):
            s = G.normal(a, 1.0) @ "s"
            G.mask(inner)(s > 0.0, s) @ "mk"
            return s
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["s"].set(0.5), upd=C["s"].set(0.3), sel="s")

    def cat_sample_shape(n):
        @G.gen
        def m(a):
            w = G.normal(a, 1.0) @ "w"
            idx = G.categorical(logits=jnp.stack([w, w * 0.0, -w]), sample_shape=n) @ "idx"
            return w
        return dict(model=m, args=lambda k: (k_float(k),), obs=C["w"].set(0.2), upd=C["w"].set(0.3), sel="w")

    def many_sites(n):
        @G.gen
        def m(a):
            prev = a
            for j in range(n):
                prev = G.normal(prev * 0.5, 1.0) @ f"x{j}"
            return prev
        return dict(model=m, args=lambda k: (k_float(k),), obs=C[f"x{n - 1}"].set(0.2), upd=C["x0"].set(0.3), sel="x0")

    def k_float(k):
        dev = G._lib.get().device
        return 0.25 if k is None else torch.linspace(-1, 1, B, device=dev)

    return [
        ("scalar sites", scalar, [1]),
        ("vector-valued site `normal(s * xs, 0.5)`", vector_site, [8, 500, 5000]),
        ("latent vector feeding the next vector site (8-schools at J)", latent_vector, [8, 40, 1000]),
        ("latent vector gathered at a table of group indices, `normal(theta[group], 0.5) @ \"y\"` (5 observations per group)", gathered, [8, 40, 200]),
        ("plate of a `@gen` element (`elem.vmap()`)", plate, [8, 100, 5000]),
        ("bare plate `normal.vmap()`, values used: `jnp.sum(vs)`, `vs[1]`", bare_plate_used, [8, 100, 5000]),
        ("`repeat` of clusters read at a traced index `means[z]`", plate_traced_index, [8, 40]),
        ("scan, stacked outputs summed in the model", scan, [8, 100]),
        ("scan with an ARRAY carry `jnp.zeros(2)`", scan_array_carry, [8, 100]),
        ("plate of long scans (T = 20)", plate_of_scans, [3, 40]),
        ("20 scans of `size` steps written directly, `step.scan(n=size).vmap()`; index = `IndexRequest(j, IndexRequest(t, Update))`",
         direct_plate_of_scans, [8, 100]),
        ("masked call `mask(inner)(flag, s)`", masked, [1]),
        ("`categorical(logits, sample_shape=n)`", cat_sample_shape, [8, 100, 5000]),
        ("static model of n sites", many_sites, [20, 100]),
    ]


def run_cell(form, method, one, probe):
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, Selection, SelectionBuilder as S, StaticRequest, Update
    from genjax_amd.inference.requests import HMC
    m, args = form["model"], form["args"](None if one else 1)
    key = lambda s: G.key(s) if one else G.split(G.key(s), B)
    sel = form["sel"]
    sel_t = sel if isinstance(sel, tuple) else (sel,)
    if method == "simulate":
        with probe:
            m.simulate(key(1), args)
        return
    if method == "importance":
        with probe:
            m.importance(key(1), form["obs"], args)
        return
    tr = m.simulate(key(1), args)            # (setup: not counted)
    if method == "hmc" and form.get("hmc") is False:
        raise LookupError("n/a")
    with probe:
        _edit_cell(form, method, one, m, args, key, tr, sel_t)


def _edit_cell(form, method, one, m, args, key, tr, sel_t):
    import genjax_amd as G
    from genjax_amd import ChoiceMapBuilder as C, Diff, IndexRequest, Regenerate, SelectionBuilder as S, StaticRequest, Update
    from genjax_amd.inference.requests import HMC
    if method == "assess":
        m.assess(tr.get_choices(), args) if not one else m.assess(tr.get_choices(), args, batch_shape=())
        return
    nd = Diff.no_change(args)
    if method == "update":
        m.update(key(2), tr, form["upd"], nd)
    elif method == "regenerate":
        Regenerate(S[sel_t]).edit(key(2), tr, nd)
    elif method == "rejuvenate":
        leaf = G.Rejuvenate(G.normal, lambda chm: (chm.get_value(), 0.5))
        req = leaf
        for comp in reversed(sel_t):
            req = StaticRequest({comp: req})
        req.edit(key(2), tr, nd)
    elif method == "hmc":
        HMC(S[form.get("sel_scalar", sel_t)] if "sel_scalar" in form else S[sel_t], 1e-3, L=2).edit(key(2), tr, nd)
    elif method == "index":
        ix = form.get("index")
        if ix is None:
            raise LookupError("n/a")
        addr, site, size = ix
        if addr is None:            # the model IS the nest of combinators: a nested request, no StaticRequest around it
            IndexRequest(min(3, size - 1), IndexRequest(2, Update(C[site].set(0.1)))).edit(key(2), tr, nd)
            return
        sub = Update(C[site].set(0.1)) if not isinstance(site, tuple) else Update(C[site + (0, "z")].set(0.1))
        StaticRequest({addr: IndexRequest(min(3, size - 1), sub)}).edit(key(2), tr, nd)


def cell(form_builder, size, method, one):
    import genjax_amd as G
    from genjax_amd.program import ProgramTooLarge
    G.clear_caches()
    try:
        form = form_builder(size)
        p = Probe()
        run_cell(form, method, one, p)
        return p.how()
    except LookupError:
        return "—"
    except ProgramTooLarge as e:
        return "**ProgramTooLarge**"
    except Exception as e:      # noqa: BLE001
        msg = " ".join(str(e).split())
        return f"**{type(e).__name__}**: {msg[:70]}"


def generate(sample=None):
    import tests.hostsim as hs
    hs.install()
    from oracle import genjax_oracle as O
    O.build()
    rows = []
    k = 0
    for name, builder, sizes in forms():
        for size in sizes:
            for one in (False, True):
                k += 1
                if sample is not None and k % sample:
                    continue
                cells = [cell(builder, size, mth, one) for mth in METHODS]
                rows.append((name, size, "ONE key" if one else f"{B} keys", cells))
    return rows


def render(rows):
    out = ["# SUPPORT.md — which model form runs under which method, and how",
           "",
           "GENERATED by `python tools/support_matrix.py` (do not edit): every cell was RUN on the CPU mirror of the C-ABI",
           "(`tests/hostsim`: the same tracer and site programs the GPU runs).  A cell says what the call took — programs",
           "launched, counted loops inside them (`OP_LOOP`), `chain` = a program cut into several launches",
           "(`program.split_graph`), `site by site` = ONE trace whose large plates / vector sites run on the launch axis",
           "(`sitewise.py`) — or the exception it raised (bold).  `—` = the method does not apply to the form.",
           "Methods: `simulate`; `importance` under the form's observation; `assess` of a simulated trace's choices; `update` of one",
           "upstream choice; `Regenerate` / `Rejuvenate` (through a `StaticRequest`) / `HMC` on the form's own latent; `IndexRequest`",
           "(`Update` of one element) into the form's plate / scan.  Sizes = elements of the plate / vector / scan (sites for the last form).",
           "`NotSupportedEditRequest … vmap.py:342-362` is the reference's own refusal: its `Vmap.edit` answers `Update` and `IndexRequest`",
           "only (`case _: raise NotImplementedError`), and its `Regenerate` handler sends every site of a static model a",
           "`Regenerate` (static.py:655-665), so a model that holds a plate takes no `Regenerate` there either.",
           "",
           "| form | size | batch | " + " | ".join(METHODS) + " |",
           "|---|---|---|" + "---|" * len(METHODS)]
    for name, size, batch, cells in rows:
        out.append(f"| {name} | {size} | {batch} | " + " | ".join(cells) + " |")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    t0 = time.time()
    rows = generate()
    text = render(rows)
    path = os.path.join(ROOT, "SUPPORT.md")
    if "--check" in sys.argv:
        same = os.path.exists(path) and open(path).read() == text
        print("SUPPORT.md is", "current" if same else "STALE")
        sys.exit(0 if same else 1)
    open(path, "w").write(text)
    print(f"wrote {path}: {len(rows)} rows in {time.time() - t0:.0f} s")
