"""Random `@gen` models against the oracle (test infrastructure).

One SPEC (a list of statements drawn from a small grammar: leaf sites, plates, scans, masked calls, masked plates,
plates of scans, scans of plates — unrolled and loop sizes mixed; rows of logits at one categorical site; values computed in the model read at a traced index (`means[z]`), slices of a long latent vector; long vector-valued sites (`vec`), a latent vector whose
values are the next vector site's parameters (`hvec`: 8-schools' shape), plates (`vplate`), plates of plates (`vplate2`)
and scans (`vscan`) of elements that hold such sites; one model in eight is long: 12 to 25 statements, a chain of
launches) is built twice, with the product (`genjax_amd`) and with the
oracle (`oracle/genjax_oracle.py`), and every GFI method is compared bit for bit under a batch of keys:
simulate (score, return value, every choice), importance under a random subset of constraints, assess of the
resulting choices, and `update` with a random subset of new constraints and randomly CHANGED arguments (the
per-particle argument, a table — or, for half of the top-level plates / scans / vector sites, a vector PER PARTICLE — a vector of flags), an `IndexRequest` into one plate / scan, `Regenerate` and MH moves
where the reference's combinators answer them.  The reference paths restated by the oracle:
static.py:255-466 (handlers), vmap.py:180-275, scan.py:200-503, mask.py:96-262."""
import numpy as np
import torch

from oracle import genjax_oracle as O

SMALL, LARGE = 3, 20          # unrolled / counted-loop sizes (the product switches at 16)


# ---------------------------------------------------------------------------
# spec
# ---------------------------------------------------------------------------
def random_spec(rng, n_stmts=None, allow_nested=True):
    kinds = ["leaf", "leaf", "call", "plate", "scan", "mask", "mplate", "mscan", "vec", "hvec", "vplate", "vscan", "bplate", "ascan"] + (["plate_of_scans", "scan_of_plates", "vplate2"] if allow_nested else [])
    stmts = []
    if n_stmts is None:
        # (one model in eight is LONG: more sites than one launch stores — a chain of launches, program.split_graph)
        n_stmts = int(rng.integers(12, 26)) if rng.random() < 0.125 else int(rng.integers(2, 5))
    for i in range(n_stmts):
        kind = kinds[int(rng.integers(len(kinds)))]
        st = dict(kind=kind, name=f"s{i}", c1=float(np.float32(rng.uniform(-1.0, 1.0))), c2=float(np.float32(rng.uniform(-0.5, 0.5))),
                  sd=float(np.float32(rng.uniform(0.5, 2.0))), src=["a", "prev"][int(rng.integers(2))])
        if kind == "leaf":
            st["dist"] = ["normal", "normal", "uniform", "flip", "beta", "categorical", "bernoulli"][int(rng.integers(7))]
            # (no draw of its own, so that the seeds' other models stay what they were) rows of logits at ONE site:
            # `categorical(logits [4, 3])` per particle, four draws under the site's key
            st["rows"] = 4 if st["dist"] == "categorical" and st["c1"] > 0.3 else 0
        if kind == "bplate":            # a plate of a BARE distribution (`normal.vmap()`), its values computed with afterwards
            st["n"] = [SMALL, LARGE, 130][int(rng.integers(3))]
        if kind == "ascan":             # a scan whose carry is a 2-vector (an ARRAY as the initial carry), outputs summed afterwards
            st["T"] = [SMALL, LARGE][int(rng.integers(2))]
        if kind in ("plate", "mplate", "plate_of_scans"):
            st["n"] = [SMALL, LARGE][int(rng.integers(2))]
            st["two"] = bool(rng.integers(2))
            st["bern"] = bool(rng.integers(2)) and kind != "plate_of_scans"
        if kind in ("scan", "scan_of_plates", "plate_of_scans", "mscan", "vscan"):
            st["T"] = [SMALL, LARGE][int(rng.integers(2))]
            st["bern"] = bool(rng.integers(2)) and kind == "scan"
        if kind == "scan_of_plates":
            st["n"] = SMALL + 1          # (not the scan's own length: a [T, n] constraint then says which axis is which)
        if kind == "plate_of_scans" and st["n"] == LARGE and st["T"] == LARGE:
            st["T"] = SMALL if rng.integers(2) else LARGE
        if kind == "mask":
            st["flag"] = ["arg", True, False][int(rng.integers(3))]
        if kind == "vec":               # ONE vector-valued site of many elements (a counted loop per particle)
            st["n"] = [24, 40, 130][int(rng.integers(3))]
        if kind == "hvec":              # a LATENT vector whose values are the next vector site's parameters (8-schools' shape)
            st["n"] = [24, 40, 130][int(rng.integers(3))]
        if kind == "vscan":             # a scan (unrolled or a loop) whose step emits a long vector (an HMM with vector observations)
            st["m"] = [24, 40][int(rng.integers(2))]
        if kind == "vplate2":           # a plate of plates of elements with a long vector site: three levels with the site's loop
            st["n"] = [SMALL, LARGE][int(rng.integers(2))]
            st["n2"] = [SMALL + 1, LARGE][int(rng.integers(2))]
            st["m"] = 24
        if kind == "vplate":            # a plate (unrolled or a loop) of elements that hold such a pair of long vector sites
            st["n"] = [SMALL, LARGE][int(rng.integers(2))]
            st["m"] = [24, 40][int(rng.integers(2))]
        stmts.append(st)
    return stmts


class PerParticle(np.ndarray):
    """an extra argument that is ONE VECTOR PER PARTICLE ([B, n]: a device tensor for the product) rather than a table"""


def _per_particle(st, B):
    """half of the plates / scans / vector sites of a model run under B > 1 keys map over (compute with) a per-particle
    vector ([B, n]; a plate of plates: [B, n, n2]) instead of a launch-uniform table (decided by the statement's own
    numbers: no draw of its own)"""
    kinds = ("plate", "scan", "vec", "hvec", "vscan", "plate_of_scans", "scan_of_plates", "vplate", "vplate2", "bplate")
    return B > 1 and st["kind"] in kinds and st["sd"] > 1.25


def spec_args(spec, rng, B):
    """(a [B] f32, per-statement extra arguments): a table per plate / scan (or a vector per particle), flags per masked
    statement"""
    a = rng.normal(size=B).astype(np.float32)
    extra = []
    for st in spec:
        k = st["kind"]
        if _per_particle(st, B):
            shape = (st["n"], st["n2"]) if k == "vplate2" else ((st["T"],) if k in ("scan", "vscan", "scan_of_plates") else (st["n"],))
            extra.append(rng.normal(size=(B,) + shape).astype(np.float32).view(PerParticle))
        elif k in ("plate", "plate_of_scans", "vec", "hvec", "vplate", "bplate"):
            extra.append(rng.normal(size=st["n"]).astype(np.float32))
        elif k == "vplate2":
            extra.append(rng.normal(size=(st["n"], st["n2"])).astype(np.float32))
        elif k in ("scan", "scan_of_plates", "vscan", "ascan"):
            extra.append(rng.normal(size=st["T"]).astype(np.float32))
        elif k == "mscan":
            extra.append(np.arange(st["T"]) < int(rng.integers(0, st["T"] + 1)))
        elif k == "mplate" and B > 1 and st["sd"] > 1.25:
            extra.append((rng.random((B, st["n"])) < 0.6).view(PerParticle))          # flags and values per particle
            extra.append(rng.normal(size=(B, st["n"])).astype(np.float32).view(PerParticle))
        elif k == "mplate":
            extra.append(rng.random(st["n"]) < 0.6)
            extra.append(rng.normal(size=st["n"]).astype(np.float32))
        elif k == "mask" and st["flag"] == "arg":
            extra.append(rng.random(B) < 0.5)
    return a, extra


# ---------------------------------------------------------------------------
# the model, for either library
# ---------------------------------------------------------------------------
def build(g, spec, lit):
    """`g`: genjax_amd or the oracle module; `lit`: a float literal in that library's arithmetic"""
    def mul_add(v, c1, c2):
        return v * lit(c1) + lit(c2)

    def make_elem(st):
        @g.gen
        def elem(shared, x):
            if st.get("bern"):
                b = g.flip(lit(0.4)) @ "b"
                v = g.normal(_where(g, b, shared + x, shared - x), lit(st["sd"])) @ "v"
            else:
                v = g.normal(shared + x, lit(st["sd"])) @ "v"
            if st.get("two"):
                u = g.normal(v * lit(st["c1"]), lit(1.5)) @ "u"
                return u
            return v
        return elem

    def make_vec_elem(st):
        @g.gen
        def elem(shared, x):
            v = g.normal(shared + x, lit(st["sd"])) @ "v"
            tab = _ramp(g, st["m"], lit)
            mean = (v * tab) if g is not O else (np.asarray(v, np.float32)[..., None] * tab).astype(np.float32)
            z = g.normal(mean, lit(1.5)) @ "z"                            # a long vector site inside the element
            loc = (z * lit(st["c1"]) + tab) if g is not O else (np.asarray(z, np.float32) * lit(st["c1"]) + tab).astype(np.float32)
            g.normal(loc, lit(st["sd"])) @ "w"                           # ... whose values the next one computes with
            return v
        return elem

    def make_step(st):
        @g.gen
        def step(c, x):
            if st.get("bern"):
                b = g.flip(lit(0.4)) @ "b"
                z = g.normal(_where(g, b, c * lit(0.5) + x, x - c), lit(st["sd"])) @ "z"
            else:
                z = g.normal(c * lit(0.5) + x, lit(st["sd"])) @ "z"
            return z, z
        return step

    def make_astep(st):
        @g.gen
        def step(c, x):
            if g is O:
                c = np.asarray(c, np.float32)
                z = g.normal(((c[..., 0] * lit(0.5)).astype(np.float32) + (c[..., 1] * lit(0.25)).astype(np.float32)).astype(np.float32) + x,
                             lit(st["sd"])) @ "z"
                full = np.broadcast_shapes(np.shape(z), c[..., 0].shape)
                return np.stack([np.broadcast_to(z, full), np.broadcast_to(c[..., 0], full)], axis=-1).astype(np.float32), z
            from genjax_amd import numpy as jnp
            z = g.normal(c[0] * lit(0.5) + c[1] * lit(0.25) + x, lit(st["sd"])) @ "z"
            return jnp.stack([z, c[0]]), z
        return step

    def make_vec_row(st):
        @g.gen
        def elem(shared, x):
            if g is O:
                s_ = np.asarray(shared, np.float32)
                while s_.ndim < np.ndim(x):
                    s_ = s_[..., None]
                shared = s_
            v = g.normal(shared + x, lit(st["sd"])) @ "v"
            tab = _ramp(g, st["m"], lit)
            mean = (v * tab) if g is not O else (np.asarray(v, np.float32)[..., None] * tab).astype(np.float32)
            g.normal(mean, lit(1.5)) @ "z"
            return v

        @g.gen
        def row(shared, xs):
            return g.Vmap(elem, in_axes=(None, 0))(shared, xs) @ "r"
        return row

    def make_vstep(st):
        @g.gen
        def step(c, x):
            z = g.normal(c * lit(0.5) + x, lit(st["sd"])) @ "z"
            tab = _ramp(g, st["m"], lit)
            loc = (z * tab) if g is not O else (np.asarray(z, np.float32)[..., None] * tab).astype(np.float32)
            g.normal(loc, lit(1.25)) @ "y"                                 # the step's long vector emission
            return z, z
        return step

    def make_mstep(st):
        @g.gen
        def mstep(x):
            z = g.normal(x * lit(0.5), lit(st["sd"])) @ "z"
            return z
        return O.MaskedIterate(mstep, False) if g is O else g.masked_iterate_final()(mstep)

    def make_call(st):
        @g.gen
        def sub(m):
            p_ = g.normal(m, lit(st["sd"])) @ "p"
            q_ = g.normal(p_ * lit(st["c1"]), lit(1.25)) @ "q"
            return p_ + q_
        return sub

    def make_inner(st):
        @g.gen
        def inner(m):
            y = g.normal(m, lit(st["sd"])) @ "y"
            return y
        return inner

    def make_scan_elem(st):
        step = make_step(st)

        @g.gen
        def elem(shared, x):
            xs = _full(g, st["T"], lit)
            cT, _ = g.Scan(step, st["T"])(shared + x, xs) @ "chain"
            return cT
        return elem

    def make_plate_step(st):
        elem = make_elem(dict(st, two=False))

        @g.gen
        def step(c, x):
            vs = g.Vmap(elem, in_axes=(None, 0))(c * lit(0.5) + x, _ramp(g, st["n"], lit)) @ "row"
            v0 = vs[..., 0] if g is O else vs[0]          # (the oracle keeps the plate axis last, behind the batch)
            z = g.normal(v0 * lit(0.25) + c * lit(0.5), lit(st["sd"])) @ "z"
            return z, z
        return step
    parts = []
    for st in spec:
        k = st["kind"]
        parts.append(dict(st, fn={"plate": make_elem, "mplate": make_elem, "scan": make_step, "mask": make_inner, "call": make_call, "mscan": make_mstep,
                                  "plate_of_scans": make_scan_elem, "scan_of_plates": make_plate_step, "vplate": make_vec_elem, "vscan": make_vstep, "vplate2": make_vec_row, "ascan": make_astep}.get(k, lambda s: None)(st)))

    @g.gen
    def model(a, *extra):
        prev = a
        it = iter(extra)
        for st in parts:
            k, name = st["kind"], st["name"]
            src = a if st["src"] == "a" else prev
            m = mul_add(src, st["c1"], st["c2"])
            if k == "leaf":
                if st["dist"] == "normal":
                    prev = g.normal(m, lit(st["sd"])) @ name
                elif st["dist"] == "uniform":
                    prev = g.uniform(m - lit(1.0), m + lit(2.0)) @ name
                elif st["dist"] == "beta":
                    p_ = g.beta(m * m + lit(1.5), lit(2.5)) @ name
                    prev = p_ * lit(2.0) + src
                elif st["dist"] == "categorical" and st.get("rows"):
                    i_ = g.categorical(logits=_stack3_rows(g, m, st["rows"], lit)) @ name
                    if st["c2"] > 0.0:          # `means[zs]`: three values computed in the model, read at the four draws
                        prev = _take3(g, m, src, i_, lit)[..., 2] if g is O else _take3(g, m, src, i_, lit)[2]
                    else:
                        prev = _where(g, (i_[..., 1] if g is O else i_[1]) == 1, m, src)
                elif st["dist"] == "categorical":
                    i_ = g.categorical(logits=_stack3(g, m, lit)) @ name
                    prev = _take3(g, m, src, i_, lit) if st["c1"] < -0.3 else _where(g, i_ == 1, m, src)       # `means[z]`
                elif st["dist"] == "bernoulli":
                    b = g.bernoulli(logits=m) @ name
                    prev = _where(g, b, src, m)
                else:
                    b = g.flip(lit(0.35)) @ name
                    prev = _where(g, b, m, src)
            elif k == "call":
                prev = st["fn"](m) @ name
            elif k == "vec":
                xs = next(it)
                mean = (m + xs * lit(st["c1"])) if g is not O else (np.asarray(m, np.float32)[..., None] + xs * lit(st["c1"])).astype(np.float32)
                g.normal(mean, lit(st["sd"])) @ name                     # (its values live in memory only)
                prev = m
            elif k == "hvec":
                xs = next(it)
                mean = (m + xs * lit(st["c1"])) if g is not O else (np.asarray(m, np.float32)[..., None] + xs * lit(st["c1"])).astype(np.float32)
                z = g.normal(mean, lit(st["sd"])) @ (name + "z")
                if st["c2"] > 0.25:         # a SLICE of the latent vector (a random walk's increments): n - 1 elements
                    loc = ((z[1:] - z[:-1]) * lit(st["c2"]) + xs[1:]) if g is not O else (
                        (np.asarray(z, np.float32)[..., 1:] - np.asarray(z, np.float32)[..., :-1]).astype(np.float32) * lit(st["c2"]) + xs[..., 1:]).astype(np.float32)
                else:
                    loc = (z * lit(st["c2"]) + xs) if g is not O else (np.asarray(z, np.float32) * lit(st["c2"]) + xs).astype(np.float32)
                g.normal(loc, lit(st["sd"])) @ (name + "y")              # the model computes with the latent vector's values
                prev = m
                if st["c1"] > 0.3:          # ... and reads ONE of them at a traced index (`mus[z]`: a search loop, engine.StepInput._read_at)
                    if g is O:
                        i_ = np.where(np.asarray(m, np.float32) > 0.0, 3, 17)
                        zz = np.asarray(z, np.float32)          # (a launch-uniform constraint is [n], the index one per particle)
                        lead = np.broadcast_shapes(zz.shape[:-1], i_.shape)
                        prev = np.take_along_axis(np.broadcast_to(zz, lead + zz.shape[-1:]), np.broadcast_to(i_, lead)[..., None], axis=-1)[..., 0]
                    else:
                        prev = z[_where(g, m > 0.0, 3, 17)]
            elif k in ("vplate", "vplate2"):
                g.Vmap(st["fn"], in_axes=(None, 0))(m, next(it)) @ name
                prev = m
            elif k == "plate":
                vs = g.Vmap(st["fn"], in_axes=(None, 0))(m, next(it)) @ name
                prev = m
                if st["c2"] > 0.0:          # the plate's RETURN values computed with (vmap.py:180-191: plain stacked arrays):
                    prev = _use_stacked(g, m, vs, lit)          # their sum, one element, one at a traced index
            elif k == "bplate":
                xs = next(it)
                locs = (m + xs * lit(st["c1"])) if g is not O else (np.asarray(m, np.float32)[..., None] + xs * lit(st["c1"])).astype(np.float32)
                if g is O and locs.ndim >= 2 and locs.shape[0] == 1:
                    locs = locs[0]          # (a value the oracle carries as [1]: the same for every particle, i.e. launch-uniform)
                vs = g.Vmap(g.normal, in_axes=(0, None))(locs, lit(st["sd"])) @ name          # `normal.vmap(in_axes=(0, None))`
                prev = _use_stacked(g, m, vs, lit)
            elif k == "ascan":
                z0 = _zeros2(g)
                cT, ys = g.Scan(st["fn"], st["T"])(z0, next(it)) @ name
                prev = _use_stacked(g, (cT[..., 0] if g is O else cT[0]) + m, ys, lit)
            elif k == "mplate":
                flags, tab = next(it), next(it)
                g.Vmap(g.MaskCombinator(st["fn"]), in_axes=(0, None, 0))(flags, m, tab) @ name
            elif k in ("scan", "vscan"):
                cT, ys_ = g.Scan(st["fn"], st["T"])(m, next(it)) @ name
                prev = cT
                if k == "scan" and st["c2"] > 0.0:          # the scan's stacked outputs computed with (scan.py:221-233)
                    prev = _use_stacked(g, cT, ys_, lit)
            elif k == "mscan":
                prev = st["fn"](m, next(it)) @ name
            elif k == "mask":
                flag = next(it) if st["flag"] == "arg" else st["flag"]
                g.MaskCombinator(st["fn"])(flag, m) @ name
            elif k == "plate_of_scans":
                g.Vmap(st["fn"], in_axes=(None, 0))(m, next(it)) @ name
                prev = m
            elif k == "scan_of_plates":
                cT, _ = g.Scan(st["fn"], st["T"])(m, next(it)) @ name
                prev = cT
        return prev
    model._fuzz_parts = parts
    return model


def _use_stacked(g, m, vs, lit):
    """m + sum(vs) / 8 + vs[1] + vs[m > 0 ? 0 : 2]: the stacked return values of a plate / the outputs of a scan, computed
    with in the model — in element order (`jnp.sum` as the build defines it inside a program)"""
    if g is O:
        vs = np.asarray(vs, np.float32)
        m = np.asarray(m, np.float32)
        acc = np.zeros(vs.shape[:-1], np.float32)
        for j in range(vs.shape[-1]):
            acc = (acc + vs[..., j]).astype(np.float32)
        i_ = np.where(m > 0.0, 0, 2)
        lead = np.broadcast_shapes(vs.shape[:-1], i_.shape)
        pick = np.take_along_axis(np.broadcast_to(vs, lead + vs.shape[-1:]), np.broadcast_to(i_, lead)[..., None], axis=-1)[..., 0]
        return (((m + (acc * lit(0.125)).astype(np.float32)).astype(np.float32) + vs[..., 1]).astype(np.float32) + pick).astype(np.float32)
    from genjax_amd import numpy as jnp
    return m + jnp.sum(vs) * lit(0.125) + vs[1] + vs[jnp.where(m > 0.0, 0, 2)]


def _zeros2(g):
    if g is O:
        return np.zeros(2, np.float32)
    from genjax_amd import numpy as jnp
    return jnp.zeros(2)


def _where(g, b, x, y):
    if g is O:
        return np.where(b, x, y).astype(np.float32)
    from genjax_amd import numpy as jnp
    return jnp.where(b, x, y)


def _stack3(g, m, lit):
    """the logits [m, 0, -m] of a three-way categorical, per particle"""
    if g is O:
        m = np.asarray(m, np.float32)
        return np.stack([m, np.zeros_like(m), (-m).astype(np.float32)], axis=-1)
    from genjax_amd import numpy as jnp
    return jnp.stack([m, m * 0.0, -m])


def _take3(g, m, src, i_, lit):
    """[m, src, m / 2][i_]: values computed in the model, read at a traced index (a draw, or a vector of draws)"""
    if g is O:
        tab = np.stack(np.broadcast_arrays(np.asarray(m, np.float32), np.asarray(src, np.float32),
                                           (np.asarray(m, np.float32) * lit(0.5)).astype(np.float32)), axis=-1)
        i_ = np.asarray(i_)
        if i_.ndim == tab.ndim - 1:
            return np.take_along_axis(tab, i_[..., None], axis=-1)[..., 0]
        return np.take_along_axis(np.broadcast_to(tab, i_.shape[:-1] + tab.shape[-1:]), i_, axis=-1)
    from genjax_amd import numpy as jnp
    return jnp.stack([m, src, m * lit(0.5)])[i_]


def _stack3_rows(g, m, J, lit):
    """J rows of such logits per particle: row j is [m r_j, 0, -m r_j] with r = 0.5 .. 1.5"""
    r = np.linspace(0.5, 1.5, J).astype(np.float32)
    if g is O:
        mr = (np.asarray(m, np.float32)[..., None] * r).astype(np.float32)
        return np.stack([mr, np.zeros_like(mr), (-mr).astype(np.float32)], axis=-1)
    from genjax_amd import numpy as jnp
    mr = m * jnp.array(r)
    return jnp.stack([mr, mr * 0.0, -mr], axis=-1)


def _full(g, T, lit):
    a = np.linspace(-0.5, 0.5, T).astype(np.float32)
    if g is O:
        return a
    from genjax_amd import numpy as jnp
    return jnp.array(a)


def _ramp(g, n, lit):
    a = np.linspace(0.0, 1.0, n).astype(np.float32)
    if g is O:
        return a
    from genjax_amd import numpy as jnp
    return jnp.array(a)


# ---------------------------------------------------------------------------
# addresses of a spec: (product path, oracle key, shape after the batch, dtype kind, masked?)
# ---------------------------------------------------------------------------
def addresses(spec):
    out = []
    for st in spec:
        k, nm = st["kind"], st["name"]
        if k == "leaf":
            kind_ = {"flip": "b", "bernoulli": "b", "uniform": "u", "beta": "u", "categorical": "i"}.get(st["dist"], "f")
            out.append(((nm,), (nm,), (st["rows"],) if st.get("rows") else (), kind_, False, st))
        elif k in ("vec", "bplate"):
            out.append(((nm,), (nm,), (st["n"],), "f", False, st))
        elif k == "ascan":
            out.append(((nm, "z"), (nm, "z"), (st["T"],), "f", False, st))
        elif k == "vplate2":
            out.append(((nm, "r", "v"), (nm, "r", "v"), (st["n"], st["n2"]), "f", False, st))
            out.append(((nm, "r", "z"), (nm, "r", "z"), (st["n"], st["n2"], st["m"]), "f", False, st))
        elif k == "vplate":
            out.append(((nm, "v"), (nm, "v"), (st["n"],), "f", False, st))
            out.append(((nm, "z"), (nm, "z"), (st["n"], st["m"]), "f", False, st))
            out.append(((nm, "w"), (nm, "w"), (st["n"], st["m"]), "f", False, st))
        elif k == "hvec":
            out.append(((nm + "z",), (nm + "z",), (st["n"],), "f", False, st))
            out.append(((nm + "y",), (nm + "y",), (st["n"] - (1 if st["c2"] > 0.25 else 0),), "f", False, st))
        elif k == "call":
            out.append(((nm, "p"), (nm, "p"), (), "f", False, st))
            out.append(((nm, "q"), (nm, "q"), (), "f", False, st))
        elif k in ("plate", "mplate"):
            if st.get("bern"):
                out.append(((nm, "b"), (nm, "b"), (st["n"],), "b", k == "mplate", st))
            out.append(((nm, "v"), (nm, "v"), (st["n"],), "f", k == "mplate", st))
            if st.get("two"):
                out.append(((nm, "u"), (nm, "u"), (st["n"],), "f", k == "mplate", st))
        elif k == "scan":
            if st.get("bern"):
                out.append(((nm, "b"), (nm, "b"), (st["T"],), "b", False, st))
            out.append(((nm, "z"), (nm, "z"), (st["T"],), "f", False, st))
        elif k == "vscan":
            out.append(((nm, "z"), (nm, "z"), (st["T"],), "f", False, st))
            out.append(((nm, "y"), (nm, "y"), (st["T"], st["m"]), "f", False, st))
        elif k == "mscan":
            out.append(((nm, "z"), (nm, "z"), (st["T"],), "f", True, st))
        elif k == "mask":
            out.append(((nm, "y"), (nm, "y"), (), "f", True, st))
        elif k == "plate_of_scans":
            out.append(((nm, "chain", "z"), (nm, "chain", "z"), (st["n"], st["T"]), "f", False, st))
        elif k == "scan_of_plates":
            out.append(((nm, "row", "v"), (nm, "row", "v"), (st["T"], st["n"]), "f", False, st))
            out.append(((nm, "z"), (nm, "z"), (st["T"],), "f", False, st))
    return out


def _g_constraint(G, cons):
    """launch-uniform values ([*shape] host arrays / Python floats) or one value per particle ([B, *shape] device tensors)"""
    from genjax_amd import ChoiceMapBuilder as C, _lib, numpy as jnp
    cm = C.n()
    for (path, _, shape, kind, _, _), val in cons:
        if isinstance(val, tuple):            # a SUBSET of a plate's / scan's elements: `C[name, idx_array, site]`
            idx, vals = val
            cm = cm | C[(path[0], idx) + tuple(path[1:])].set(vals)
            continue
        key = (path[0],) + (slice(None),) * len(shape) + tuple(path[1:]) if shape else path
        if val.ndim > len(shape):
            v = torch.from_numpy(np.ascontiguousarray(val)).to(_lib.get().device)
        else:
            v = jnp.array(val) if shape else float(val)
        cm = cm | C[key].set(v)
    return cm


def _o_constraint(cons):
    if not cons:
        return O.ChoiceMap()
    out = {}
    for (_, okey, shape, _, _, _), val in cons:
        if isinstance(val, tuple):
            out[okey] = O.indexed(val[1], val[0], shape[0])
        else:
            out[okey] = val if val.ndim else np.float32(val)
    return O.C.d(out)


def _pick_constraints(spec, rng, p, B):
    cons = []
    for ad in addresses(spec):
        path, okey, shape, kind, masked, st = ad
        if kind != "f" or rng.random() > p:
            continue
        form = int(rng.integers(3))
        if form == 2 and len(shape) == 1 and not masked and st["kind"] not in ("vec", "hvec", "bplate"):
            m_ = int(rng.integers(1, min(3, shape[0]) + 1))
            idx = np.sort(rng.choice(shape[0], size=m_, replace=False))
            cons.append((ad, (idx, rng.normal(size=m_).astype(np.float32))))
            continue
        cons.append((ad, rng.normal(size=((B,) if form == 1 else ()) + tuple(shape)).astype(np.float32)))
    return cons


# ---------------------------------------------------------------------------
# one comparison
# ---------------------------------------------------------------------------
def _np(v):
    return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def _choice(chm, path):
    from genjax_amd.core.mask import Mask
    v = chm[path]
    return (_np(v.value), _np(v.flag)) if isinstance(v, Mask) else (_np(v), None)


def _ochoice(chm, okey):
    v = chm[okey]
    return (np.asarray(v.value), np.asarray(v.flag)) if isinstance(v, O.Mask) else (np.asarray(v), None)


def _same_choices(spec, tr, otr, B, what):
    ch, och = tr.get_choices(), otr.get_choices()
    for path, okey, shape, kind, masked, st in addresses(spec):
        v, f = _choice(ch, path)
        ov, of = _ochoice(och, okey)
        if st["kind"] == "vscan" and len(shape) == 2 and ov.ndim >= 2 and ov.shape[-2:] == (shape[1], shape[0]):
            ov = np.moveaxis(ov, -1, -2)          # (the oracle's scans stack what they sample along a LAST axis: [.., m, T])
        assert np.array_equal(*np.broadcast_arrays(v, ov)), (what, path)      # (a launch-uniform constraint stays a scalar)
        assert (f is None) == (of is None), (what, path, "masked on one side only")
        if f is not None:
            full = np.broadcast_shapes(v.shape, ov.shape, f.shape, of.shape)
            lead = lambda x: x.reshape(x.shape + (1,) * (len(full) - x.ndim))
            assert np.array_equal(np.broadcast_to(lead(f), full), np.broadcast_to(lead(of), full)), (what, path, "flags")


class OverTheLimits(Exception):
    """the drawn model does not fit one site program (64 launch slots, 8 tables, 64 live values)"""


def run_one(seed, B=7, allow_nested=True, verbose=False):
    from genjax_amd.program import ProgramTooLarge
    try:
        return _run_one(seed, B, allow_nested, verbose)
    except ProgramTooLarge as e:
        raise OverTheLimits(str(e)) from None
    except ValueError as e:
        if "exceeds the ABI slot limits" in str(e):
            raise OverTheLimits(str(e)) from None
        raise


def _run_one(seed, B=7, allow_nested=True, verbose=False):
    import genjax_amd as G
    from genjax_amd import Diff, _lib, numpy as jnp
    rng = np.random.default_rng(seed)
    spec = random_spec(rng, allow_nested=allow_nested)
    if verbose:
        print(seed, [(s["kind"], s.get("n"), s.get("T"), s.get("dist"), s.get("flag")) for s in spec])
    dev = _lib.get().device
    model, omodel = build(G, spec, float), build(O, spec, np.float32)
    a, extra = spec_args(spec, rng, B)

    def g_args(a_, extra_):
        out = [torch.from_numpy(a_).to(dev)]
        for e in extra_:
            out.append(torch.from_numpy(np.ascontiguousarray(e)).to(dev) if isinstance(e, PerParticle) or (
                e.shape[:1] == (B,) and e.dtype == bool and e.ndim == 1 and not _is_table(e, spec, B)) else jnp.array(e))
        return tuple(out)
    k, ok = G.split(G.key(seed), B), O.split(O.key(seed), B)
    tr, otr = model.simulate(k, g_args(a, extra)), omodel.simulate(ok, (a,) + tuple(extra))
    assert np.array_equal(_np(tr.get_score()), otr.get_score()), (seed, "simulate score")
    assert np.array_equal(_np(tr.get_retval()), np.broadcast_to(otr.get_retval(), (B,))), (seed, "simulate retval")
    _same_choices(spec, tr, otr, B, (seed, "simulate"))
    cons = _pick_constraints(spec, rng, 0.5, B)
    tri, w = model.importance(k, _g_constraint(G, cons), g_args(a, extra))
    otri, ow = omodel.importance(ok, _o_constraint(cons), (a,) + tuple(extra))
    assert np.array_equal(_np(w), np.broadcast_to(ow, (B,))), (seed, "importance weight")
    assert np.array_equal(_np(tri.get_score()), otri.get_score()), (seed, "importance score")
    _same_choices(spec, tri, otri, B, (seed, "importance"))
    s_, _ = model.assess(tri.get_choices(), g_args(a, extra))
    assert np.array_equal(_np(s_), otri.get_score()), (seed, "assess")
    # update: new constraints, changed arguments
    cons2 = _pick_constraints(spec, rng, 0.35, B)
    a2, extra2 = spec_args(spec, rng, B)
    change = [bool(rng.integers(2)) for _ in range(1 + len(extra))]
    new_a = a2 if change[0] else a
    new_extra = [e2 if c else e for e, e2, c in zip(extra, extra2, change[1:])]
    ga_old, ga_new = g_args(a, extra), g_args(new_a, new_extra)
    diffs = tuple(Diff(n_, G.UnknownChange) if c else Diff.no_change(o_) for o_, n_, c in zip(ga_old, ga_new, change))
    k2, ok2 = G.split(G.key(seed + 1000), B), O.split(O.key(seed + 1000), B)
    new, wu, _, bwd = model.update(k2, tri, _g_constraint(G, cons2), diffs)
    onew, owu, odis = omodel.update(ok2, otri, _o_constraint(cons2), (new_a,) + tuple(new_extra))
    assert np.array_equal(_np(wu), np.broadcast_to(owu, (B,))), (seed, "update weight", change, [c[0][0] for c in cons2])
    assert np.array_equal(_np(new.get_score()), onew.get_score()), (seed, "update score")
    _same_choices(spec, new, onew, B, (seed, "update"))
    for (path, okey, shape, kind, masked, st), _v in cons2:       # the discard: the old values of what was constrained
        if st["kind"] in ("scan", "scan_of_plates", "plate_of_scans", "mscan", "vscan", "ascan") or isinstance(_v, tuple):
            continue                                   # (the oracle restates no discard for scans; a subset's is masked)
        d, _f = _choice(bwd, path)
        od, _of = _ochoice(odis, okey)
        assert np.array_equal(*np.broadcast_arrays(d, od)), (seed, "discard", path)
    # IndexRequest on one plate / scan of the model (vmap.py:277-332, scan.py:325-416) through a StaticRequest
    targets = [st for st in spec if st["kind"] in ("plate", "scan", "vplate", "vscan")]
    if targets:
        from genjax_amd import ChoiceMapBuilder as C, IndexRequest, StaticRequest, Update
        st = targets[int(rng.integers(len(targets)))]
        platelike = st["kind"] in ("plate", "vplate")
        size = st["n"] if platelike else st["T"]
        idx = int(rng.integers(size))
        site = "v" if platelike else "z"
        val = np.float32(rng.normal())
        o_elem = next(p_ for p_ in _parts_of(omodel) if p_["name"] == st["name"])["fn"]

        class _OIdx:
            def edit(self, kk, subtrace, gen_fn, args_):
                sub = O.C.d({(site,): val})
                if platelike:
                    at = (args_[0], np.asarray(args_[1])[..., idx])
                    return O.vmap_edit_index(gen_fn, kk, subtrace, idx, lambda k_, sl, a_: o_elem.update(k_, sl, sub, a_)[:2], at)
                return O.scan_edit_index(gen_fn, kk, subtrace, args_, idx, lambda k_, sl, a_: o_elem.update(k_, sl, sub, a_)[:2])
        per_particle = st["kind"] in ("plate", "scan") and bool(rng.integers(2))
        if per_particle:                      # one index per particle (a traced idx under the reference's vmap)
            idx_host = rng.integers(size, size=B).astype(np.int32)

            class _OIdx:                       # noqa: F811
                def edit(self, kk, subtrace, gen_fn, args_):
                    sub = O.C.d({(site,): val})
                    if not platelike:
                        return O.scan_edit_index_per_particle(gen_fn, kk, subtrace, args_, idx_host,
                                                              lambda k_, sl, a_: o_elem.update(k_, sl, sub, a_)[:2])
                    return O.vmap_edit_index_per_particle(
                        gen_fn, kk, subtrace, idx_host, lambda k_, sl, a_: o_elem.update(k_, sl, sub, a_)[:2],
                        lambda j: (args_[0], np.asarray(args_[1])[..., j]))
            idx = torch.from_numpy(idx_host).to(dev)
        k4, ok4 = G.split(G.key(seed + 3000), B), O.split(O.key(seed + 3000), B)
        req = StaticRequest({st["name"]: IndexRequest(idx, Update(C[site].set(float(val))))})
        ix, wx, _, _ = req.edit(k4, tri, Diff.no_change(ga_old))
        oix, owx = omodel.edit_static(ok4, otri, {st["name"]: _OIdx(), (st["name"],): _OIdx()}, (a,) + tuple(extra))
        assert np.array_equal(_np(wx), np.broadcast_to(owx, (B,))), (seed, "index request weight", st["kind"], size, per_particle)
        assert np.array_equal(_np(ix.get_score()), oix.get_score()), (seed, "index request score")
        _same_choices(spec, ix, oix, B, (seed, "index request"))
    # regenerate a random selection (Vmap.edit answers Update and IndexRequest only, vmap.py:342-362: models without plates)
    if all(st["kind"] in ("leaf", "scan", "call") for st in spec):
        from genjax_amd import Regenerate, SelectionBuilder as S, static
        picked = [ad for ad in addresses(spec) if rng.random() < 0.5]
        if picked:
            sel = None
            for path, *_ in picked:
                sel = S[path] if sel is None else sel | S[path]
            k3, ok3 = G.split(G.key(seed + 2000), B), O.split(O.key(seed + 2000), B)
            rg, wr, _, _ = Regenerate(sel).edit(k3, tri, Diff.no_change(ga_old))
            org, owr, _ = omodel.regenerate(ok3, otri, O.selection(*[okey for _, okey, *_ in picked]), (a,) + tuple(extra))
            assert np.array_equal(_np(wr), np.broadcast_to(owr, (B,))), (seed, "regenerate weight")
            assert np.array_equal(_np(rg.get_score()), org.get_score()), (seed, "regenerate score")
            _same_choices(spec, rg, org, B, (seed, "regenerate"))
            # ... and as ONE fused Metropolis-Hastings move per particle (static.run_mh: propose, accept, select)
            mh, acc, wm = static.run_mh(model, G.split(G.key(seed + 4000), B), tri, Regenerate(sel), Diff.no_change(ga_old))
            osel = O.selection(*[okey for _, okey, *_ in picked])
            omh, oacc, owm = O.rejuvenate(O.key(seed + 4000), otri,
                                          lambda k_, tr_: omodel.regenerate(k_, tr_, osel, (a,) + tuple(extra))[:2])
            assert np.array_equal(_np(acc), oacc), (seed, "MH accept")
            assert np.array_equal(_np(wm), np.broadcast_to(owm, (B,))), (seed, "MH weight")
            assert np.array_equal(_np(mh.get_score()), omh.get_score()), (seed, "MH score")
            _same_choices(spec, mh, omh, B, (seed, "MH"))
    leaves = [st for st in spec if st["kind"] == "leaf" and st["dist"] == "normal"]
    if leaves and all(st["kind"] in ("leaf", "scan", "call") for st in spec):
        from genjax_amd import Rejuvenate, StaticRequest, static
        st = leaves[int(rng.integers(len(leaves)))]
        sd = float(np.float32(rng.uniform(0.2, 1.0)))
        req = StaticRequest({st["name"]: Rejuvenate(G.normal, lambda chm: (chm.get_value(), sd))})
        oreq = {st["name"]: O.Rejuvenate(O.normal, lambda chm: (chm.get_value(), np.float32(sd)))}
        mh, acc, wm = static.run_mh(model, G.split(G.key(seed + 5000), B), tri, req, Diff.no_change(ga_old))
        omh, oacc, owm = O.rejuvenate(O.key(seed + 5000), otri,
                                      lambda k_, tr_: omodel.edit_static(k_, tr_, oreq, (a,) + tuple(extra)))
        assert np.array_equal(_np(acc), oacc), (seed, "Rejuvenate accept")
        assert np.array_equal(_np(wm), np.broadcast_to(owm, (B,))), (seed, "Rejuvenate weight")
        assert np.array_equal(_np(mh.get_score()), omh.get_score()), (seed, "Rejuvenate score")
        _same_choices(spec, mh, omh, B, (seed, "Rejuvenate"))
    return spec


def _parts_of(model):
    return model._fuzz_parts


def _is_table(e, spec, B):
    """a boolean argument of length B that is a plate's flag TABLE (a masked plate of B elements), not per-particle flags"""
    return any(st["kind"] == "mplate" and st["n"] == B for st in spec)


def run_smc_one(seed, K=33):
    """`ImportanceK(Target(model, args, constraints), K).run_smc(key)` (ref smc.py:298-315) of a random model under ONE
    key: the K log-weights, the particles' scores and choices, the log-marginal-likelihood estimate (smc.py:95-96)."""
    from genjax_amd.program import ProgramTooLarge
    try:
        return _run_smc_one(seed, K)
    except ProgramTooLarge as e:
        raise OverTheLimits(str(e)) from None
    except ValueError as e:
        if "exceeds the ABI slot limits" in str(e):
            raise OverTheLimits(str(e)) from None
        raise


def _run_smc_one(seed, K):
    import genjax_amd as G
    from genjax_amd import numpy as jnp
    from genjax_amd.inference.smc import ImportanceK
    rng = np.random.default_rng(seed)
    spec = [st for st in random_spec(rng) if not (st["kind"] == "mask" and st["flag"] == "arg")]     # (shared arguments only)
    if not spec:
        return None
    model, omodel = build(G, spec, float), build(O, spec, np.float32)
    a, extra = spec_args(spec, rng, 1)
    a0 = np.float32(a[0])
    cons = [c for c in _pick_constraints(spec, rng, 0.5, 1) if isinstance(c[1], tuple) or c[1].ndim == len(c[0][2])]
    g_args = (float(a0),) + tuple(jnp.array(e) for e in extra)
    tgt = G.Target(model, g_args, _g_constraint(G, cons))
    otgt = O.Target(omodel, (a0,) + tuple(extra), _o_constraint(cons))
    coll = ImportanceK(tgt, k_particles=K).run_smc(G.key(seed))
    ocoll = O.ImportanceK(otgt, K).run_smc(O.key(seed))
    assert np.array_equal(_np(coll.get_log_weights()), ocoll.get_log_weights()), (seed, "log weights")
    assert np.array_equal(_np(coll.get_particles().get_score()), ocoll.get_particles().get_score()), (seed, "scores")
    _same_choices(spec, coll.get_particles(), ocoll.get_particles(), K, (seed, "particles"))
    # (the weights are equal bit for bit; their logsumexp is a fixed f32 tree in the product, held to the f64 value —
    #  the oracle's own is a sequential f32 sum, 1e-4 off by 2^18 terms)
    olw = np.asarray(ocoll.get_log_weights(), np.float64)
    olml = float(np.log(np.sum(np.exp(olw - olw.max()))) + olw.max() - np.log(K))
    lml = float(_np(coll.get_log_marginal_likelihood_estimate()))
    assert abs(lml - olml) <= 4e-6 * max(1.0, abs(olml)), (seed, "log ML", lml, olml)
    # ChangeTarget to the same model under OTHER constraints (smc.py:370-396: new weight - old score + old log-weight)
    from genjax_amd.inference.smc import ChangeTarget
    cons_b = [c for c in _pick_constraints(spec, rng, 0.5, 1) if isinstance(c[1], tuple) or c[1].ndim == len(c[0][2])]
    # (the FIRST target may constrain a SUBSET of a plate's / scan's elements — `C[name, idx_array, site]`: the selection
    #  of such an `Indexed` layer selects nothing below it (choice_map.py:1494-1496, 658-663), so `filter_to_unconstrained`
    #  keeps the WHOLE site among the latents, the listed elements with the values they were constrained to)
    ct = ChangeTarget(ImportanceK(tgt, k_particles=K), G.Target(model, g_args, _g_constraint(G, cons_b))).run_smc(G.key(seed + 13))
    oct_ = O.ChangeTarget(O.ImportanceK(otgt, K), O.Target(omodel, (a0,) + tuple(extra), _o_constraint(cons_b))).run_smc(O.key(seed + 13))
    assert np.array_equal(_np(ct.get_log_weights()), oct_.get_log_weights()), (seed, "ChangeTarget log weights")
    assert np.array_equal(_np(ct.get_particles().get_score()), oct_.get_particles().get_score()), (seed, "ChangeTarget scores")
    _same_choices(spec, ct.get_particles(), oct_.get_particles(), K, (seed, "ChangeTarget particles"))
    # the same algorithm under a BATCH of keys (the reference's vmap over run_smc): [keys, K] log-weights
    if K <= 64:
        colls = ImportanceK(tgt, k_particles=K).run_smc(G.split(G.key(seed + 11), 3))
        ocolls = O.ImportanceK(otgt, K).run_smc(O.split(O.key(seed + 11), 3))
        assert np.array_equal(_np(colls.get_log_weights()), ocolls.get_log_weights()), (seed, "log weights under a batch of keys")
    # resample the collection (a random scheme): the ancestors, and every leaf of the structured traces gathered
    from genjax_amd.inference import smc
    kind = ["systematic", "stratified", "multinomial"][int(rng.integers(3))]
    res = smc.resample(G.key(seed + 7), coll, kind)
    cdf, total, M, shift = O.weight_cdf(ocoll.get_log_weights())
    anc = O.ancestors({"systematic": O.SYSTEMATIC, "stratified": O.STRATIFIED, "multinomial": O.MULTINOMIAL}[kind], O.key(seed + 7), cdf)
    assert np.array_equal(_np(res.ancestors), anc), (seed, "ancestors", kind)
    ores = O.gather_trace(ocoll.get_particles(), anc)
    assert np.array_equal(_np(res.get_particles().get_score()), ores.get_score()), (seed, "resampled scores")
    _same_choices(spec, res.get_particles(), ores, K, (seed, "resampled particles"))
    # ChangeTarget to OTHER ARGUMENTS as well (the tables a plate / scan / vector site runs over replaced: data arriving)
    a3, extra3 = spec_args(spec, rng, 1)
    g_args3 = (float(np.float32(a3[0])),) + tuple(jnp.array(e) for e in extra3)
    ct3 = ChangeTarget(ImportanceK(tgt, k_particles=K), G.Target(model, g_args3, _g_constraint(G, cons_b))).run_smc(G.key(seed + 17))
    oct3 = O.ChangeTarget(O.ImportanceK(otgt, K), O.Target(omodel, (np.float32(a3[0]),) + tuple(extra3), _o_constraint(cons_b))).run_smc(O.key(seed + 17))
    assert np.array_equal(_np(ct3.get_log_weights()), oct3.get_log_weights()), (seed, "ChangeTarget to other arguments: log weights")
    assert np.array_equal(_np(ct3.get_particles().get_score()), oct3.get_particles().get_score()), (seed, "ChangeTarget to other arguments: scores")
    return spec


def run_big_one(seed, n_big=4099, K=65):
    """a random model whose LAST statement is a plate of thousands of elements: ONE trace of it (an unbatched key: the
    site-by-site path with the plate on the launch axis, sitewise.py) and `ImportanceK` over K particles (the plate
    deferred: run after the program over particles x elements, combinators.Vmap._defer) — against the oracle"""
    from genjax_amd.program import ProgramTooLarge
    try:
        return _run_big_one(seed, n_big, K)
    except ProgramTooLarge as e:
        raise OverTheLimits(str(e)) from None
    except ValueError as e:
        if "exceeds the ABI slot limits" in str(e):
            raise OverTheLimits(str(e)) from None
        raise


def _run_big_one(seed, n_big, K):
    import genjax_amd as G
    from genjax_amd import Diff, numpy as jnp
    from genjax_amd.inference.smc import ImportanceK
    rng = np.random.default_rng(seed)
    head = [st for st in random_spec(rng, n_stmts=int(rng.integers(1, 3)), allow_nested=False)
            if st["kind"] in ("leaf", "call", "scan") and st.get("T", SMALL) == SMALL]
    last = dict(kind="plate", name="big", c1=0.5, c2=0.1, sd=float(np.float32(rng.uniform(0.5, 2.0))), src="prev", n=n_big,
                two=bool(rng.integers(2)), bern=bool(rng.integers(2)))
    spec = head + [last]
    model, omodel = build(G, spec, float), build(O, spec, np.float32)
    a, extra = spec_args(spec, rng, 1)
    a0 = np.float32(a[0])
    g_args, o_args = (float(a0),) + tuple(jnp.array(e) for e in extra), (a0,) + tuple(extra)
    obs = rng.normal(size=n_big).astype(np.float32)
    site = "u" if last["two"] else "v"
    cons = [(next(ad for ad in addresses(spec) if ad[0] == ("big", site)), obs)]
    # ONE trace
    tr, otr = model.simulate(G.key(seed), g_args), omodel.simulate(O.key(seed), o_args)
    assert np.array_equal(_np(tr.get_score()), otr.get_score()), (seed, "one trace: simulate score")
    _same_choices(spec, tr, otr, 1, (seed, "one trace: simulate"))
    tri, w = model.importance(G.key(seed + 1), _g_constraint(G, cons), g_args)
    otri, ow = omodel.importance(O.key(seed + 1), _o_constraint(cons), o_args)
    assert np.array_equal(_np(w), ow) and np.array_equal(_np(tri.get_score()), otri.get_score()), (seed, "one trace: importance")
    a2, extra2 = spec_args(spec, rng, 1)
    new_args = (float(a2[0]),) + g_args[1:]
    new, wu, _, _ = model.update(G.key(seed + 2), tri, _g_constraint(G, []), (Diff(new_args[0], G.UnknownChange),) +
                                 tuple(Diff.no_change(x) for x in g_args[1:]))
    onew, owu, _ = omodel.update(O.key(seed + 2), otri, O.ChoiceMap(), (np.float32(a2[0]),) + o_args[1:])
    assert np.array_equal(_np(wu), owu) and np.array_equal(_np(new.get_score()), onew.get_score()), (seed, "one trace: update")
    # K particles under one key: the plate deferred
    coll = ImportanceK(G.Target(model, g_args, _g_constraint(G, cons)), k_particles=K).run_smc(G.key(seed + 3))
    ocoll = O.ImportanceK(O.Target(omodel, o_args, _o_constraint(cons)), K).run_smc(O.key(seed + 3))
    assert np.array_equal(_np(coll.get_log_weights()), ocoll.get_log_weights()), (seed, "K particles: log weights")
    assert np.array_equal(_np(coll.get_particles().get_score()), ocoll.get_particles().get_score()), (seed, "K particles: scores")
    return spec
