"""Distributions as leaf generative functions.

Each distribution supplies the SYMBOLIC sampler / log-density (program ops that
run inside the fused kernel: csrc/gmx_dist.h) and inherits the leaf GFI
semantics of the reference's `Distribution` / `ExactDensity`
(src/genjax/_src/generative_functions/distributions/distribution.py:90-419);
the instances mirror the hot `tfp_distribution` wrappers
(distributions/tensorflow_probability/__init__.py: normal :259, uniform :294,
beta :82, flip :155, bernoulli :72, categorical :102-104).

Vector-valued sites use ONE site key; element j takes counter j and the
array-valued log_prob is summed into one site score (distribution.py:383-396).
"""
from __future__ import annotations

import warnings

import numpy as np

from . import tracer as T
from .core.choice_map import ChoiceMap
from .core.generative import GenerativeFunction
from .tracer import Expr, current_graph


def _obj(x):
    return isinstance(x, np.ndarray) and x.dtype == object


def _bcast(args):
    """Broadcast symbolic args; returns (list of flat element tuples, shape)."""
    arrs = [a if isinstance(a, np.ndarray) else np.asarray(a, dtype=object) for a in args]
    if all(a.ndim == 0 for a in arrs):
        return [tuple(a.item() for a in arrs)], ()
    b = np.broadcast_arrays(*arrs)
    shape = b[0].shape
    return [tuple(x[idx] for x in b) for idx in np.ndindex(shape)], shape


def _seq_sum(terms):
    acc = terms[0]
    for t in terms[1:]:
        acc = acc + t
    return acc


def _traced_call(v, args) -> bool:
    """True inside a traced function when a value / parameter is symbolic: the call contributes to the program being
    traced instead of launching one of its own."""
    if not T.is_tracing():
        return False
    import dataclasses

    def sym(x):
        if T.is_symbolic(x):
            return True
        if isinstance(x, (tuple, list)):
            return any(sym(y) for y in x)
        if dataclasses.is_dataclass(x) and not isinstance(x, type):
            return any(sym(getattr(x, f_.name)) for f_ in dataclasses.fields(x))
        return False
    return sym(v) or sym(tuple(args))


class Distribution(GenerativeFunction):
    name = "distribution"
    value_dtype = "f32"
    sample_op = None       # program op of the scalar sampler
    logpdf_op = None
    n_params = 2
    param_names: tuple = ()

    # -- argument handling ---------------------------------------------------
    def handle_kwargs(self):
        return self          # exact_density types reply with self (distribution.py:468)

    def canon(self, args) -> tuple:
        """Accept (a, b) or the kwargs form ((a, b), {kw}) (distribution.py:448-466).  `sample_shape=` (a tuple, an
        int or a Const of either; tensorflow_probability/__init__.py:52-55): the site draws `sample_shape + batch`
        values from its one key — element j of the row-major array takes counter j — and its score is their sum;
        carried by broadcasting the first parameter to that shape."""
        if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple):
            pos, kw = args
            kw = dict(kw)
            ss = kw.pop("sample_shape", None)
            out = self.from_kwargs(tuple(pos), kw)
            if ss is not None:
                ss = ss.unwrap() if hasattr(ss, "unwrap") else ss
                ss = (int(ss),) if isinstance(ss, (int, np.integer)) else tuple(int(x) for x in ss)
                if ss and out:
                    first = out[0] if isinstance(out[0], np.ndarray) else np.asarray(out[0], dtype=object)
                    out = (np.broadcast_to(first, ss + first.shape),) + tuple(out[1:])
            return out
        return tuple(args)

    def from_kwargs(self, pos, kw):
        out = list(pos)
        for name in self.param_names[len(pos):]:
            if name not in kw:
                raise TypeError(f"{self.name}: missing argument {name!r}")
            out.append(kw.pop(name))
        if kw:
            raise TypeError(f"{self.name}: unexpected keyword arguments {sorted(kw)}")
        return tuple(out)

    # -- symbolic sampler / density -------------------------------------------------
    def _conv_value(self, v):
        return {"f32": T.as_float, "i32": T.as_int, "bool": T.as_bool}[self.value_dtype](v)

    def sym_sample(self, key: Expr, args: tuple):
        if self.sample_op is None and type(self).random_weighted is not Distribution.random_weighted:
            # a user-defined Distribution (distribution.py:90-106): its own `random_weighted(key, *args)`, traced —
            # e.g. composed from the built-in distributions' random_weighted, which answer symbolically to a traced key
            return self.random_weighted(key, *args)[1]
        g = current_graph()
        elems, shape = _bcast(args)
        out = []
        if getattr(g, "elem_from_index", False):
            # the ELEMENTS of one vector-valued site on the launch axis (sitewise.vector_site): launch element i is the
            # site's element i — counter = the global index, from the one site key
            from .program import ELEM_INDEX
            if shape != () or self.sample_op is None:
                raise NotImplementedError(f"{self.name}: a large vector-valued site takes scalar elements")
            ops = tuple(T.as_float(x).node for x in elems[0])
            return Expr(g.add(self.sample_op, (key.node,) + ops, imm=ELEM_INDEX, dtype=self.value_dtype))
        hoist = g.__dict__.get("noise_hoist") if self.sample_op in ("S_NORMAL", "S_UNIFORM") else None
        for e, a in enumerate(elems):
            if hoist is not None:
                # noise-ahead (engine.NoiseHoist): the standard-normal / unit-uniform draw comes from memory — a
                # background program drew it from the same key — and only the sampler's own last operations stay
                # here, in its order (csrc/gmx_dist.h gmx_normal_sample: z * scale + loc; gmx_uniform_sample:
                # lo + (hi - lo) * u)
                z = hoist.request(key.node, e, "normal" if self.sample_op == "S_NORMAL" else "uniform")
                if z is not None and self.sample_op == "S_NORMAL":
                    out.append(z * T.as_float(a[1]) + T.as_float(a[0]))
                    continue
                if z is not None:
                    lo, hi = T.as_float(a[0]), T.as_float(a[1])
                    out.append(lo + (hi - lo) * z)
                    continue
            ops = tuple(T.as_float(x).node for x in a)
            out.append(Expr(g.add(self.sample_op, (key.node,) + ops, imm=e, dtype=self.value_dtype)))
        if shape == ():
            return out[0]
        arr = np.empty(len(out), dtype=object)
        arr[:] = out
        return arr.reshape(shape)

    def sym_logpdf(self, v, args: tuple) -> Expr:
        """estimate_logpdf: elementwise log_prob summed in element order."""
        if self.logpdf_op is None and type(self).estimate_logpdf is not Distribution.estimate_logpdf:
            return T.as_float(self.estimate_logpdf(None, v, *args))          # a user-defined Distribution
        g = current_graph()
        elems, _ = _bcast((v,) + tuple(args))
        terms = []
        for el in elems:
            x = self._conv_value(el[0])
            ops = tuple(T.as_float(a).node for a in el[1:])
            terms.append(Expr(g.add(self.logpdf_op, (x.node,) + ops, dtype="f32")))
        return _seq_sum(terms)

    # -- direct (non-traced) convenience: tfp-style sample / logpdf ---------------------
    def sample(self, key, *args, **kwargs):
        a = self.canon((args, kwargs)) if kwargs else args
        return self.simulate(key, a).get_retval()

    def logpdf(self, v, *args, **kwargs):
        a = self.canon((args, kwargs)) if kwargs else args
        return self.assess(ChoiceMap.choice(v), a)[0]

    # -- GFI: answered by the static engine with a one-site program ----------------------
    def simulate(self, key, args):
        from .static import run_gfi
        return run_gfi(self, "simulate", key, args)

    def generate(self, key, constraint, args):
        from .static import run_gfi
        return run_gfi(self, "generate", key, args, constraint=constraint)

    def assess(self, sample, args, batch_shape=None):
        if _traced_call(sample.get_value() if isinstance(sample, ChoiceMap) else sample, args):
            v = sample.get_value() if isinstance(sample, ChoiceMap) else sample      # inside a traced function
            return self.sym_logpdf(v, self.canon(tuple(args))), v
        from .static import run_gfi
        return run_gfi(self, "assess", None, args, constraint=sample, batch_shape=batch_shape)

    def edit(self, key, trace, edit_request, argdiffs):
        from .static import run_edit
        return run_edit(self, key, trace, edit_request, argdiffs)

    def project(self, key, trace, selection):
        import torch
        s = trace.get_score()
        return s if selection.check() else (torch.zeros_like(s) if hasattr(s, "shape") else 0.0)

    def random_weighted(self, key, *args):
        if isinstance(key, Expr):                  # a traced key: inside a user-defined distribution / @gen function
            a = self.canon(tuple(args))
            v = self.sym_sample(key, a)
            return self.sym_logpdf(v, a), v
        tr = self.simulate(key, args)
        return tr.get_score(), tr.get_retval()

    def estimate_logpdf(self, key, v, *args):
        if _traced_call(v, args):
            return self.sym_logpdf(v, self.canon(tuple(args)))
        return self.assess(ChoiceMap.choice(v), args)[0]

    def __repr__(self):
        return f"genjax.{self.name}"


class _Normal(Distribution):
    name, sample_op, logpdf_op = "normal", "S_NORMAL", "L_NORMAL"
    param_names = ("loc", "scale")


class _Uniform(Distribution):
    name, sample_op, logpdf_op = "uniform", "S_UNIFORM", "L_UNIFORM"
    param_names = ("low", "high")


class _Beta(Distribution):
    name, sample_op, logpdf_op = "beta", "S_BETA", "L_BETA"
    param_names = ("concentration1", "concentration0")


class _Flip(Distribution):
    """Bernoulli(probs=p, dtype=bool)"""
    name, sample_op, logpdf_op = "flip", "S_FLIP", "L_FLIP"
    value_dtype = "bool"
    param_names = ("p",)


class _Bernoulli(Distribution):
    """Bernoulli(logits=...) — a bare parameter means logits and warns
    (implicit_logit_warning, distribution.py:479-500)."""
    name, sample_op, logpdf_op = "bernoulli", "S_BERNL", "L_BERNL"
    value_dtype = "i32"
    param_names = ("logits",)

    def canon(self, args):
        if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple):
            pos, kw = args
            kw = dict(kw)
            kw.pop("sample_shape", None)
            if "probs" in kw:
                from . import numpy as jnp
                p = kw.pop("probs")
                return (jnp.log(p) - jnp.log1p(-p),) if T.is_symbolic(p) or T.is_tracing() else (_logit_host(p),)
            if "logits" in kw:
                return (kw.pop("logits"),)
            args = pos
        if len(args) == 1:
            warnings.warn("The use of a bare argument to genjax.bernoulli is deprecated. Please specify "
                          "`logits=` or `probs=`. The default, which will be used in this case, is logits.",
                          DeprecationWarning, stacklevel=3)
        return tuple(args)


def _logit_host(p):
    p = np.asarray(p, dtype=np.float32)
    return np.log(p) - np.log1p(-p)


class _Categorical(Distribution):
    """Categorical over the LAST axis of logits (bare / logits=) or log(probs=).
    Sampling is Gumbel-max with gumbel counter = category index, first max wins
    (SURVEY.md App. A.3); log_prob(k) = logits[k] - logsumexp(logits)."""
    name = "categorical"
    value_dtype = "i32"
    param_names = ("logits",)

    def canon(self, args):
        shape = None
        if len(args) == 2 and isinstance(args[1], dict) and isinstance(args[0], tuple):
            pos, kw = args
            kw = dict(kw)
            shape = kw.pop("sample_shape", None)
            if "probs" in kw:
                from . import numpy as jnp
                p = kw.pop("probs")
                if T.is_tracing():
                    p = np.asarray(p, dtype=object)          # (rows of probabilities keep their shape: J draws at the site)
                    flat = np.empty(p.size, dtype=object)
                    flat[:] = [T.lift(x) for x in p.reshape(-1)]
                    return self._shaped((jnp.log(flat.reshape(p.shape)),), shape)
                return self._shaped((np.log(np.asarray(p, dtype=np.float32)),), shape)
            if "logits" in kw:
                return self._shaped((kw.pop("logits"),), shape)
            args = pos
        if len(args) == 1 and not isinstance(args[0], dict):
            warnings.warn("The use of a bare argument to genjax.categorical is deprecated. Please specify "
                          "`logits=` or `probs=`. The default, which will be used in this case, is logits.",
                          DeprecationWarning, stacklevel=3)
        return self._shaped(tuple(args), shape)

    def simulate(self, key, args):
        """`categorical.simulate(key, (logits[n, K],))` under ONE key (7_application_dirichlet_mixture_model.ipynb c10,
        update_datapoint_assignment): n draws, row i / category k on gumbel counter i * K + k — what
        `jax.random.categorical(key, logits)` of that shape does.  From 65 rows on, one GPU thread per row."""
        import torch
        from .sitewise import VECTOR_SITE_MIN, sum_defined
        l = args[0] if len(args) == 1 else None
        if key is not None and tuple(key.shape) == () and isinstance(l, torch.Tensor) and l.ndim == 2 \
                and l.shape[0] >= VECTOR_SITE_MIN:
            from .static import DistributionTrace, run_gfi
            tr = run_gfi(self, "simulate", key, (l,), batch_shape=(int(l.shape[0]),), elem_index=True)
            out = DistributionTrace(self, tuple(args), tr.value, sum_defined(tr.score))
            out._elem_scores = tr.score
            return out
        return super().simulate(key, args)

    @staticmethod
    def _shaped(args, shape):
        """`sample_shape=n` (tfp sample_n): n draws from the same logits at ONE site; draw j, category k
        takes gumbel counter j*K + k.  Unrolled, so only for small n (plates of data belong in
        inference.gibbs / the particle axis)."""
        shape = shape.unwrap() if hasattr(shape, "unwrap") else shape          # a Const[int] (the reference passes one)
        if shape is None or shape == ():
            return args
        n = int(shape[0] if isinstance(shape, (tuple, list)) else shape)
        return args + (("sample_shape", n),)

    SAMPLE_SHAPE_LOOP_MIN = 17      # draws of one site from which they run as a counted loop (static._vector_site_loop)
    SPILL_MIN = 24                  # computed logits beyond this many are read from memory inside that loop

    def loop_site(self, args):
        """`categorical(logits, sample_shape=n)` with MANY draws at one site (the n assignments of
        7_application_dirichlet_mixture_model.ipynb c6) as the body of ONE counted loop per particle: draw j takes the
        gumbel counters j * K .. j * K + K - 1 of the one site key, as the unrolled form does; its log-probability is
        logits[idx_j] - logsumexp(logits), added in draw order.  Returns (n, sample(key, t), logpdf(x, t)) — built
        BEFORE the loop opens (the logits and their normaliser are loop-invariant) — or None for a site that stays
        unrolled."""
        n = args[1][1] if len(args) == 2 and isinstance(args[1], tuple) and args[1][:1] == ("sample_shape",) else None
        if n is None or n < self.SAMPLE_SHAPE_LOOP_MIN:
            return None
        from . import numpy as jnp
        g = current_graph()
        l = args[0]
        long = T._long_vector(l)
        if not long and (isinstance(l, np.ndarray) and l.ndim != 1):
            return None
        ls = [T.as_float(T._elem(l, k)) for k in range(long)] if long else self._logits(args[:1])
        K = len(ls)
        lse = jnp.logsumexp(np.asarray(ls, dtype=object))
        mem = None
        if not long and K > self.SPILL_MIN and not g.loop_counts and getattr(g, "_tracing", None) is not None:
            # many logits COMPUTED in registers (jnp.log of a Dirichlet draw of 64 weights): K values alive across the
            # loop would not fit a launch — they go to memory once (Tracing.spill_vector) and each use is one load
            mem = g._tracing.spill_vector(ls)

        def sample(key, t):
            base = T.as_int(t) * K
            state = None
            for k, lk in enumerate(ls):
                lk = mem.read_in_loop(k) if mem is not None else lk
                state = g.add("S_CATSTEP", (state, key.node, lk.node, (base + k).node), imm=k, dtype="cat")
            return Expr(g.add("CATIDX", (state,), dtype="i32"))

        def logpdf(x, t):
            vi = T.as_int(x)
            if long:
                return T.as_float(T._elem(l, vi)) - lse          # one read of the logits at the drawn index
            if mem is not None:
                return mem[vi] - lse
            picked = ls[0]
            for k in range(1, K):
                picked = T.where(vi == k, ls[k], picked)
            return picked - lse
        return n, sample, logpdf

    ROWS_MAX = 64      # rows of logits at one site under a batch of keys: unrolled

    def _logits(self, args):
        l = args[0]
        l = l if isinstance(l, np.ndarray) else np.asarray(l, dtype=object)
        if l.ndim != 1:
            raise NotImplementedError("categorical: logits must be a vector per particle")
        return [T.as_float(x) for x in l]

    def _rows(self, args):
        """logits [J, K] per particle: J draws at ONE site, row j / category k on gumbel counter j * K + k (what
        `jax.random.categorical(key, logits)` of that shape does).  None for one row."""
        l = args[0]
        l = l if isinstance(l, np.ndarray) else np.asarray(l, dtype=object)
        if l.ndim != 2:
            return None
        if l.shape[0] > self.ROWS_MAX:
            raise NotImplementedError(f"categorical: {l.shape[0]} rows of logits at one site under a batch of keys are "
                                      f"unrolled (<= {self.ROWS_MAX}); write the rows as a plate (`categorical.vmap()`), "
                                      "or run ONE trace (sitewise.py)")
        return [[T.as_float(x) for x in row] for row in l]

    def sym_sample(self, key: Expr, args: tuple):
        g = current_graph()
        n = args[1][1] if len(args) == 2 and isinstance(args[1], tuple) and args[1][:1] == ("sample_shape",) else None
        rows = self._rows(args[:1]) if n is None else None
        if rows is not None:
            if getattr(g, "elem_from_index", False):
                raise NotImplementedError("categorical: rows of logits under a site whose elements ride the launch axis")
            arr = np.empty(len(rows), dtype=object)
            for j, ls in enumerate(rows):
                state = None
                for k, lk in enumerate(ls):
                    state = g.add("S_CATSTEP", (state, key.node, lk.node, g.const_i32(j * len(ls) + k)), imm=k, dtype="cat")
                arr[j] = Expr(g.add("CATIDX", (state,), dtype="i32"))
            return arr
        ls = self._logits(args[:1])
        out = []
        row = None
        if getattr(g, "elem_from_index", False):      # draw i of `sample_shape = n` on the launch axis: counters i * K + k
            if n is not None:
                raise NotImplementedError("categorical: the elements of a large sample_shape are scalar draws")
            row = Expr(g.add("LDIDX", dtype="i32")) * len(ls)
        for j in range(n or 1):
            state = None
            for k, lk in enumerate(ls):
                ctr = g.const_i32(j * len(ls) + k) if row is None else (row + k).node
                state = g.add("S_CATSTEP", (state, key.node, lk.node, ctr), imm=k, dtype="cat")
            # the index lives in the second register of the state pair
            out.append(Expr(g.add("CATIDX", (state,), dtype="i32")))
        if n is None:
            return out[0]
        arr = np.empty(n, dtype=object)
        arr[:] = out
        return arr

    def sym_logpdf(self, v, args: tuple) -> Expr:
        from . import numpy as jnp
        rows = self._rows(args[:1]) if len(args) == 1 else None
        if rows is not None:                        # one value per row of logits: the rows' log-probabilities, summed
            vs = np.broadcast_to(np.asarray(v, dtype=object), (len(rows),)) if not isinstance(v, np.ndarray) else v.reshape(-1)
            if len(vs) != len(rows):
                raise ValueError(f"categorical: {len(vs)} values for {len(rows)} rows of logits")
            return _seq_sum([self.sym_logpdf(vj, (np.asarray(ls, dtype=object),)) for vj, ls in zip(vs, rows)])
        ls = self._logits(args[:1])
        lse = jnp.logsumexp(np.asarray(ls, dtype=object))
        if isinstance(v, np.ndarray):               # sample_shape draws: sum of the per-draw log-probabilities
            terms = []
            for vj in v.reshape(-1):
                vi = T.as_int(vj)
                picked = ls[0]
                for k in range(1, len(ls)):
                    picked = T.where(vi == k, ls[k], picked)
                terms.append(picked - lse)
            return _seq_sum(terms)
        if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
            return ls[int(v)] - lse                 # a category fixed at trace time (enumeration)
        vi = T.as_int(v)
        picked = ls[0]
        for k in range(1, len(ls)):
            picked = T.where(vi == k, ls[k], picked)
        return picked - lse


class _LogNormal(Distribution):
    """tfd.LogNormal(loc, scale): exp of a Normal draw; log_prob(x) = Normal.log_prob(log x) - log x."""
    name = "log_normal"
    param_names = ("loc", "scale")

    def sym_sample(self, key: Expr, args: tuple):
        from . import numpy as jnp
        return jnp.exp(normal.sym_sample(key, args))

    def sym_logpdf(self, v, args: tuple) -> Expr:
        from . import numpy as jnp
        elems, _ = _bcast((v,) + tuple(args))
        terms = []
        for x, loc, scale in elems:
            lx = jnp.log(T.as_float(x))
            terms.append(normal.sym_logpdf(lx, (loc, scale)) - lx)
        return _seq_sum(terms)


class _HalfNormal(Distribution):
    """tfd.HalfNormal(scale): |z * scale|; log_prob(x) = -0.5 (x/scale)^2 - (log scale + 0.5 log(pi/2)), x >= 0."""
    name = "half_normal"
    param_names = ("scale",)

    def sym_sample(self, key: Expr, args: tuple):
        from . import numpy as jnp
        scale = args[0]
        z = normal.sym_sample(key, (scale * 0.0 if isinstance(scale, np.ndarray) else 0.0, 1.0)) \
            if not isinstance(scale, np.ndarray) else normal.sym_sample(key, (np.full(scale.shape, 0.0, dtype=object), 1.0))
        return jnp.abs(z * scale)

    def sym_logpdf(self, v, args: tuple) -> Expr:
        from . import numpy as jnp
        import math
        elems, _ = _bcast((v, args[0]))
        terms = []
        for x, scale in elems:
            x, scale = T.as_float(x), T.as_float(scale)
            r = x / scale
            lp = (r * r) * -0.5 - (jnp.log(scale) + 0.5 * math.log(math.pi / 2.0))
            terms.append(T.where(x < 0.0, float("-inf"), lp))
        return _seq_sum(terms)


class _Exponential(Distribution):
    """tfd.Exponential(rate) (tensorflow_probability/__init__.py:150): -log(U) / rate with U uniform on [tiny, 1) — TFP's
    sampler (exponential.py `_sample_n`) — and log_prob(x) = log rate - rate x for x >= 0.  Built from the uniform sampler and
    elementary functions (no kernel op of its own), like log_normal / half_normal."""
    name = "exponential"
    param_names = ("rate",)
    TINY = float(np.finfo(np.float32).tiny)

    def sym_sample(self, key: Expr, args: tuple):
        from . import numpy as jnp
        rate = args[0]
        lo = np.full(rate.shape, self.TINY, dtype=object) if isinstance(rate, np.ndarray) else self.TINY
        u = uniform.sym_sample(key, (lo, 1.0))
        return -jnp.log(u) / rate

    def sym_logpdf(self, v, args: tuple) -> Expr:
        from . import numpy as jnp
        elems, _ = _bcast((v, args[0]))
        terms = []
        for x, rate in elems:
            x, rate = T.as_float(x), T.as_float(rate)
            terms.append(T.where(x < 0.0, float("-inf"), jnp.log(rate) - rate * x))
        return _seq_sum(terms)


class _Dirichlet(Distribution):
    """Dirichlet(concentration) over the LAST axis (tfp/__init__.py:125).  TFP draws log-space Gammas and
    normalises: x = exp(lg - logsumexp(lg)); log_prob = sum xlogy(a - 1, x) - lbeta(a).  The Gamma
    stream is the build's (element k draws from split(site key)[k]; PARITY UNPINNED, as for Beta)."""
    name = "dirichlet"
    param_names = ("concentration",)

    def _conc(self, args):
        a = args[0]
        a = a if isinstance(a, np.ndarray) else np.asarray(a, dtype=object)
        if a.ndim != 1:
            raise NotImplementedError("dirichlet: concentration must be a vector per particle")
        return [T.as_float(x) for x in a]

    def sym_sample(self, key: Expr, args: tuple):
        from . import numpy as jnp
        g = current_graph()
        lg = np.empty(len(self._conc(args)), dtype=object)
        for k, ak in enumerate(self._conc(args)):
            lg[k] = Expr(g.add("S_LOGGAMMA", (key.node, ak.node), imm=k, dtype="f32"))
        return jnp.exp(lg - jnp.logsumexp(lg))

    def sym_logpdf(self, v, args: tuple) -> Expr:
        from . import numpy as jnp
        al = self._conc(args)
        x = v if isinstance(v, np.ndarray) else np.asarray(v, dtype=object)
        # (each term is added where it is computed — the sums of _seq_sum, in program order: 64 components keep ONE
        #  register, not 64 until the end)
        acc = None
        for ak, xk in zip(al, x.reshape(-1)):
            am = ak - 1.0
            term = T.where(am == 0.0, 0.0, am * jnp.log(T.as_float(xk)))      # xlogy(a - 1, x)
            acc = term if acc is None else acc + term
        lgs = None
        for ak in al:
            lg = jnp.lgamma(ak)
            lgs = lg if lgs is None else lgs + lg
        lbeta = lgs - jnp.lgamma(_seq_sum(al))
        return acc - lbeta


def exact_density(sample, logpdf, name="exact_density"):
    """`genjax.exact_density(sampler, logpdf)` (distribution.py:529-560).  Both functions are TRACED into
    the site program: `logpdf(v, *args)` with genjax_amd.numpy ops, `sample(key, *args)` by composing the
    symbolic samplers of the built-in distributions (`genjax_amd.normal.sym_sample(key, (loc, scale))`)."""
    class _Custom(Distribution):
        def sym_sample(self, key, args):
            return sample(key, *args)

        def sym_logpdf(self, v, args):
            return T.as_float(logpdf(v, *args))
    _Custom.name = name
    return _Custom()


def tfp_distribution(*_a, **_k):
    raise NotImplementedError("tfp_distribution wraps tensorflow_probability, which this stack does not use; "
                              "define the distribution with genjax_amd.exact_density")


class _HalfCauchy(Distribution):
    name = "half_cauchy"

    def sym_sample(self, key, args):
        raise NotImplementedError("half_cauchy: not on the hot path (SURVEY App. C)")
    sym_logpdf = sym_sample


normal = _Normal()
uniform = _Uniform()
beta = _Beta()
flip = _Flip()
bernoulli = _Bernoulli()
categorical = _Categorical()
dirichlet = _Dirichlet()
log_normal = _LogNormal()
half_normal = _HalfNormal()
exponential = _Exponential()
half_cauchy = _HalfCauchy()
