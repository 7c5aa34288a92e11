"""Particle ensembles sharded across the GPUs of one node.

One process per GPU (`torch.distributed`; backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Rank g owns the contiguous block of global
particle indices [g*n, (g+1)*n).  Nothing about the result depends on the
number of ranks:

  * per-particle keys are derived from the GLOBAL index (GMX_KEY_SPLIT with
    index_offset = g*n);
  * weights are exact integers in block floating point over tiles of 1024 consecutive GLOBAL indices
    (include/genmi.h "Resampling"): two numbers per tile (max log-weight, fixed-point weight sum) determine the
    whole integer CDF — shards therefore start on a tile boundary (n % 1024 == 0 when sharded);
  * ONE all-gather of the ranks' tile-statistics blocks (12 bytes per 1024 particles) gives every rank the global
    max, hence the global exponent, and EVERY rank's integer total (gmx_shard_totals) — exact integer sums: any
    partition gives the same CDF;
  * rank r resolves the output slots that fall into ITS mass interval
    [offset_r, offset_r + total_r) — for systematic / stratified resampling a
    contiguous slot range [S_r, E_r) every rank derives from the totals alone.
    Slots it owns itself become ancestor indices for the next step's fused gather; the states for slots other
    ranks own go into fixed-capacity send blocks (gmx_shard_step_tiles, which rebuilds the shard's CDF per tile in
    registers) and ONE equal-split all-to-all delivers them behind the receiver's local states
    (balanced weights keep all but O(sqrt(n)) particles per boundary rank-local).

So per SMC step: 1 all-gather (12 B per 1024 particles per rank) and 1 all-to-all (world * capacity * 4 B per
rank), all enqueued on the stream: the host never waits for the device inside a sweep.  (n > 2^21 per rank or > 64
ranks — `cdf_form=True` forces it: all-reduce MAX, gmx_weight_cdf against the global max, all-gather of the 8-byte
totals, gmx_shard_step, all-to-all.)  The evidence terms and the capacity
overflow flag are read once at the end; an overflow (weights so unbalanced that
a rank must ship more than `capacity` particles to one peer) re-runs the sweep
with capacity = n, which always suffices.  xGMI is point-to-point, so the
all-to-all maps onto direct peer links rather than a ring.
"""
from __future__ import annotations

import math
import os
from ctypes import c_uint32

import numpy as np
import torch

from .. import _lib
from ..core.choice_map import ChoiceMap
from ..random import Key, fold_in, lazy_split, split
from ..engine import Gathered
from .smc import MULTINOMIAL_SORTED, STRATIFIED, SYSTEMATIC, _NoiseAhead, cdf_reference, cdf_shift


def systematic_slot_bounds(offsets, total: int, n_total: int, u0: int):
    """f(c) = #{ j in [0, n_total) : (j*2^23 + u0) * total < c * n_total * 2^23 } for each
    CDF offset c — exact Python-integer arithmetic (the same predicate k_offspring /
    k_ancestors evaluate with 128-bit integers)."""
    out = []
    for c in offsets:
        c = int(c)
        if c <= 0 or total == 0:
            out.append(0)
            continue
        num = c * n_total * (1 << 23) - u0 * total          # j * 2^23 * total < num
        if num <= 0:
            out.append(0)
            continue
        den = (1 << 23) * total
        f = (num + den - 1) // den                          # ceil
        out.append(int(min(max(f, 0), n_total)))
    return out


CDF_TILE = 1024


def _check_shard_alignment(n_per_rank: int, world: int):
    """The integer CDF is defined on tiles of 1024 consecutive GLOBAL particle indices (include/genmi.h,
    "Resampling"): a shard must start on a tile boundary or the result would depend on the rank count."""
    if world > 1 and n_per_rank % CDF_TILE != 0:
        raise ValueError(f"particles per rank must be a multiple of {CDF_TILE} when sharded over {world} ranks "
                         f"(got {n_per_rank}): shards start on a tile boundary of the global CDF; "
                         f"use {((n_per_rank + CDF_TILE - 1) // CDF_TILE) * CDF_TILE}")


class ShardedBootstrapSweep(_NoiseAhead):
    """smc.BootstrapSweep over `dist.get_world_size()` ranks, n particles per rank.

    Per step (the tile-statistics form): site program -> all-gather of the statistics -> gmx_shard_step_fused (the
    totals, the slot bounds, the routing: one launch) -> all-to-all.  NOISE AHEAD as on one GPU (smc._NoiseAhead): the
    step's normal draws come from background programs on a second stream — keyed by the GLOBAL particle index, so the
    draws are the single-process ones — which matters more here than on one GPU: the chain of a sharded step is
    mostly launch boundaries and collective latency, during which the vector ALUs would idle."""

    fuse_mh = False

    def __init__(self, init, step, n_per_rank: int, T: int, dist, obs_addr="y", step_extra=None, specialize=True,
                 resample="systematic", capacity=None, always_communicate=False, rejuvenate=None, state_addr="x",
                 noise_ahead=None, cdf_form=False, fused=True, comm=None, fuse_step=None, chain_mh=True):
        """cdf_form=True: the three-collective CDF-array form (what n > 2^21 per rank or > 64 ranks take) instead of
        the tile statistics; fused=False: gmx_shard_totals + gmx_shard_step_tiles as two launches (what a vector
        state / the MH move's second leaf take) instead of gmx_shard_step_fused."""
        from .smc import _KINDS
        self.init, self.step, self.n, self.T, self.dist = init, step, int(n_per_rank), int(T), dist
        self.obs_addr = obs_addr
        self.step_extra = step_extra or (lambda t: ())
        self.specialize = specialize
        self.kind = _KINDS[resample] if isinstance(resample, str) else int(resample)
        if self.kind not in (SYSTEMATIC, STRATIFIED, MULTINOMIAL_SORTED):
            raise NotImplementedError("ShardedBootstrapSweep: the router takes the ORDERED schemes — systematic, stratified, "
                                      "multinomial_sorted (one-GPU sweeps also offer multinomial and multinomial_tiled)")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.N = self.n * self.world
        _check_shard_alignment(self.n, self.world)
        # particles one rank may ship to ONE peer per step before the slow path kicks in
        # default n/32: the slot boundaries of balanced ranks move by O(sqrt(n)) particles per step, so
        # this is a ~10x margin at n = 1e6 while the all-to-all stays ~125 KB per peer
        # (the same at world size 1, where nothing ever crosses ranks: `bench.py --sharded` then issues the collectives
        # with the message sizes a real multi-GPU step has, not a degenerate n-element all-to-all)
        self.capacity = int(capacity) if capacity else max(4096, self.n // 32)
        self.capacity = max(1, min(self.capacity, self.n))
        self.reruns = 0
        # issue the collectives even at world size 1 (exercises / times the RCCL calls on one GPU)
        self.comm = self.world > 1 or bool(always_communicate)
        self.cx = comm           # a communicator to share (comm.make_comm); None: made in prepare()
        self._own_cx = comm is None
        self.graph = None
        self._finished = True    # nothing launched yet: finish() / log_ml() / state() have nothing to wait for
        # rejuvenate: the MH request of smc.BootstrapSweep(rejuvenate=...) (BASELINE config 3).  The move on
        # a resampled particle needs the particle AND the state it was extended from, so two leaves are
        # routed (two gmx_shard_step launches and two all-to-alls per step instead of one).
        self.rejuvenate, self.state_addr = rejuvenate, state_addr
        self.noise_ahead_req = noise_ahead
        # the sorted multinomial routes against the order-statistics table of all N slots (gmx_shard_step_sorted): the
        # CDF-array form, two routing launches
        self.cdf_form, self.fused_req = bool(cdf_form) or self.kind == MULTINOMIAL_SORTED, bool(fused)
        self._noise_offset, self._noise_total = self.rank * self.n, self.N
        self.fuse_sh_req, self.fuse_sh = fuse_step, False      # None: one launch per step where it applies (prepare)
        self.chain_mh_req, self.chain_mh = bool(chain_mh), False   # rejuvenate=: the move and the extension as ONE program

    fuse_mh = False          # (the noise-ahead mixin: the chained MH + extension program's move draws hang off k_mh)

    def _chain_prog(self, t):
        if t >= 1 and self.rejuvenate is not None and getattr(self, "p_mhvm_step", None) is not None and \
                (self.chain_mh or self.fuse_mh):
            return self.p_mhvm_init if t == 1 else self.p_mhvm_step
        return self.p_init if t == 0 else self.p_step

    def _chain_step(self, t, skip_vm=False):
        self._step(t)

    def prepare(self, key: Key, ys: torch.Tensor):
        from ..static import MinimalGenerate as _MG
        be = _lib.get()
        n, T, dev, W = self.n, self.T, be.device, self.world
        # noise ahead: on request, or by default on a device with streams (specialised programs)
        want_na = self.noise_ahead_req
        # (with an MH move: the chained move + extension program as ONE launch per step — the fused peer exchange — takes
        #  the move's draws from the background stream, as the single-GPU sweep does; any other MH form draws in place)
        mh_na_ok = bool(self.rejuvenate is None or (self.chain_mh_req and be.uses_streams and self.specialize
                                                    and not self.__dict__.get("_mh_na_failed")))
        if want_na is None:
            want_na = (os.environ.get("GENMI_NOISE_AHEAD", "1") != "0" and be.uses_streams and self.specialize
                       and mh_na_ok)
        if want_na and not mh_na_ok:
            raise NotImplementedError("ShardedBootstrapSweep(noise_ahead=True, rejuvenate=...): needs the chained move + "
                                      "extension program as the one-launch sharded step (the fused peer exchange, "
                                      "specialised programs)")
        self.noise_ahead = False
        self.fuse_mh = False
        self._noise_progs = {}
        self.__dict__.pop("_noise_run_cache", None)
        if getattr(self, "graph", None) is not None:     # prepared again: the graph captured for the previous run goes
            if be.uses_streams:
                torch.cuda.synchronize()
            be.c.gmx_graph_destroy(self.graph)
            self.graph = None

        def MinimalGenerate(*a):
            return _MG(*a, hoist_noise=bool(want_na))
        self.key = key
        self.ys = ys.to(dev).float().contiguous()
        self.lw = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.cdf = torch.zeros((n,), dtype=torch.int64, device=dev)
        self.maxs = torch.zeros((T,), dtype=torch.float32, device=dev)           # global max per step
        self.totals = torch.zeros((T,), dtype=torch.int64, device=dev)            # global integer total per step
        self.total_d = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.totals_all = None      # (allocated below, once the communicator exists: a collective's destination)
        self.plan = torch.zeros((int(be.c.gmx_shard_plan_words(W)),), dtype=torch.int64, device=dev)
        self.verdict = torch.zeros((1,), dtype=torch.int64, device=dev)        # gmx_sweep_verdict: 0 fine, 1 overflow, 2 failed
        self.ws = torch.zeros(((be.c.gmx_weight_cdf_workspace(n) + 7) // 8,), dtype=torch.int64, device=dev)
        self.shift = cdf_shift(self.N)
        if self.comm and self.cx is None:
            from .comm import make_comm
            self.cx = make_comm(self.dist, dev)
        self.totals_all = self.cx.alloc((W,), torch.int64) if self.cx is not None else \
            torch.zeros((W,), dtype=torch.int64, device=dev)
        # the fused peer exchange (comm.PeerComm): no collective launch per step — tile-statistics form only
        from .smc import FUSED_RESAMPLE_MAX as _FRM
        self.peer_mode = bool(getattr(self.cx, "fused", False) and self.comm and self.kind in (SYSTEMATIC, STRATIFIED)
                              and n <= _FRM and W <= 64 and not self.cdf_form)
        obs0 = ChoiceMap.empty().set(self.obs_addr, self.ys[0])
        self.p_init = MinimalGenerate(self.init, (), obs0, (n,))
        # the state is the model's return value: a float scalar, or ONE vector of D floats per particle kept
        # struct-of-arrays ([D, n + W*C]) so every component is one routed 4-byte leaf
        if self.p_init.ro[0] != "out":
            raise NotImplementedError("ShardedBootstrapSweep: the step model must return one array (scalar or vector state)")
        dt, event, _slots = self.p_init.comp.outputs[self.p_init.ro[1]]
        if dt != "f32" or len(event) > 1:
            raise NotImplementedError("ShardedBootstrapSweep: the state must be a float scalar or a float vector")
        self.event = tuple(event)
        self.D = int(event[0]) if event else 1
        self._alloc_exchange()
        g = Gathered(self._src(0), self.idx)
        if self.rejuvenate is None:
            self.p_step = MinimalGenerate(self.step, (g,) + tuple(self.step_extra(1)), obs0, (n,))
        else:
            from ..static import MinimalMH
            self.accept = torch.zeros((n,), dtype=torch.bool, device=dev)
            self.p_step = MinimalGenerate(self.step, (self._asrc(0, local=True),) + tuple(self.step_extra(1)), obs0, (n,))
            ch = obs0.set(self.state_addr, g)
            self.p_mh_init = MinimalMH(self.init, (), ch, self.rejuvenate, (n,))
            self.p_mh_step = MinimalMH(self.step, (Gathered(self._asrc(0), self.idx),) + tuple(self.step_extra(1)), ch,
                                       self.rejuvenate, (n,))
            # the move and the extension that follows it as ONE program (static.MinimalMHGenerate, what the single-GPU
            # sweep launches per step): with the routing of the previous step as its prologue a sharded MH step is ONE
            # launch — used when that form applies (fuse_sh below), else the separate programs above
            self.p_mhvm_init = self.p_mhvm_step = None
            if self.chain_mh_req and be.uses_streams and self.specialize:
                from ..static import MinimalMHGenerate
                from .smc import BootstrapSweep as _BS
                ex1 = tuple(self.step_extra(1))
                hn = tuple(_BS.NOISE_ROOTS_MH.split(",")) if want_na else False      # the move's proposal + accept draws
                self.p_mhvm_init = MinimalMHGenerate(self.init, (), ch, self.rejuvenate, self.step, ex1, obs0, (n,),
                                                     hoist_noise=hn)
                self.p_mhvm_step = MinimalMHGenerate(self.step, (Gathered(self._asrc(0), self.idx),) + ex1, ch,
                                                     self.rejuvenate, self.step, ex1, obs0, (n,), hoist_noise=hn)
        if want_na and self.rejuvenate is not None:
            if self.p_mhvm_step is not None and self.p_mhvm_step.noise:
                self.fuse_mh = True
                self._noise_setup((self.p_init, self.p_mhvm_init, self.p_mhvm_step), n, T, dev)
            else:
                self._mh_na_failed = True
                self.noise_ahead_req = False if self.noise_ahead_req is None else self.noise_ahead_req
                return self.prepare(key, ys)
        elif want_na and self.p_step.noise:
            self._noise_setup((self.p_init, self.p_step, self.p_step), n, T, dev)
        elif want_na and self.p_init.noise:
            self.noise_ahead_req = False          # the steady-state program draws nothing ahead: the plain programs
            return self.prepare(key, ys)
        # ONE launch per sharded step: the program that gathers routes the previous step itself
        # (gmx_run_args.sh: its workgroup's tile of gmx_shard_step_peer, ancestors as tagged words) — the fused peer
        # exchange, systematic resampling, a scalar / short vector state without an MH move, a table that fits one
        # workgroup's registers, indices that fit an ancestor word
        tiles_ = (n + CDF_TILE - 1) // CDF_TILE
        # (with an MH move the program that GATHERS is the move's — it routes step t - 1 first, two leaves per state
        #  component: x_{t-1} and what it was extended from — and the extension that follows reads the moved state locally:
        #  two launches per step instead of three; round 6)
        leaves_routed = self.D * (2 if self.rejuvenate is not None else 1)
        want_fuse_sh = bool(self.peer_mode and be.uses_streams and self.specialize and self.kind == SYSTEMATIC
                            and W <= 8 and W * tiles_ <= 1024 and leaves_routed <= _lib.PEER_MAX_LEAVES
                            and n + W * self.capacity <= (1 << _lib.ANC_TAG_SHIFT) and self.fuse_sh_req is not False)
        chained = self.rejuvenate is not None and self.p_mhvm_step is not None
        self._routers = (self.p_step,) if self.rejuvenate is None else \
            ((self.p_mhvm_init, self.p_mhvm_step) if chained else (self.p_mh_init, self.p_mh_step))
        if want_fuse_sh and not any(p_.comp.is_specialized() for p_ in self._routers):
            for p_ in self._routers:
                p_.comp.set_fuse_shard_step()
        if self.specialize and be.uses_streams:
            self.p_init.comp.specialize()
            self.p_step.comp.specialize()
            if self.rejuvenate is not None:
                self.p_mh_init.comp.specialize()
                self.p_mh_step.comp.specialize()
                if chained:
                    self.p_mhvm_init.comp.specialize()
                    self.p_mhvm_step.comp.specialize()
        self.partials = torch.zeros((2, (n + 255) // 256), dtype=torch.float32, device=dev)
        # two collectives per step instead of three: the ranks all-gather their CDF TILE STATISTICS (12 bytes per
        # 1024 particles; written by the site program itself when it can, else by gmx_tile_stats), from which
        # every rank derives the global max and all the totals — no max all-reduce, no local CDF array
        from .smc import FUSED_RESAMPLE_MAX
        self.tiles_mode = (self.kind in (SYSTEMATIC, STRATIFIED) and n <= FUSED_RESAMPLE_MAX and W <= 64
                           and not self.cdf_form)
        if self.tiles_mode:
            sb = int(be.c.gmx_shard_stats_bytes(n))
            tiles = (n + CDF_TILE - 1) // CDF_TILE
            pad = tiles + (tiles & 1)
            self.stats_own = torch.zeros((sb,), dtype=torch.uint8, device=dev)
            self.stats_all = self.cx.alloc((W * sb,), torch.uint8) if self.cx is not None else \
                torch.zeros((W * sb,), dtype=torch.uint8, device=dev)
            self.tile_agg = self.stats_own[:pad * 8].view(torch.int64)
            self.tile_max = self.stats_own[pad * 8:].view(torch.float32)
            # (the one-launch step reads step t - 1's log-weights / statistics while it writes step t's: two sets)
            self.stats_own_pp = [self.stats_own, torch.zeros_like(self.stats_own)]
            self.tile_agg_pp = [b_[:pad * 8].view(torch.int64) for b_ in self.stats_own_pp]
            self.tile_max_pp = [b_[pad * 8:].view(torch.float32) for b_ in self.stats_own_pp]
        writers = (self.p_init,) + (self._routers if chained else (self.p_step,))      # who leaves a step's tile statistics
        self.fuse_sh = bool(want_fuse_sh and self.tiles_mode and all(p_.comp.fuses_shard_step() for p_ in self._routers)
                            and all(p_.comp.writes_tile_stats() for p_ in writers)
                            and all(p_.comp.resident_particles() >= n for p_ in self._routers))
        self.chain_mh = bool(chained and self.fuse_sh)
        if self.rejuvenate is not None and self.noise_ahead and not self.chain_mh:
            # the one-launch chained form did not come about (no fused peer exchange, a program that does not fit): the
            # separate programs draw in place
            self._mh_na_failed = True
            if self.noise_ahead_req:
                raise NotImplementedError("ShardedBootstrapSweep(noise_ahead=True, rejuvenate=...): the chained move + "
                                          "extension program does not run as the one-launch sharded step here")
            return self.prepare(key, ys)
        if self.fuse_sh_req and not self.fuse_sh:
            raise NotImplementedError("ShardedBootstrapSweep(fuse_step=True): needs the fused peer exchange, systematic "
                                      "resampling, specialised programs that leave tile statistics, world <= 8")
        self.lw_pp = [self.lw, torch.zeros_like(self.lw)] if self.fuse_sh else [self.lw, self.lw]
        if self.fuse_sh:
            self.sh_status = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.step_keys = []
        for t in range(T):
            ks = split(fold_in(key, t), 3)
            self.step_keys.append((ks[0], ks[1], ks[2]))
        self.sorted_tab = self.sorted_keys = None
        if self.kind == MULTINOMIAL_SORTED:
            # every rank draws the SAME table of the N global slots from the step's resampling key (integers)
            if self.N >= 1 << 31:
                raise NotImplementedError("resample='multinomial_sorted' across ranks: n_per_rank * world < 2^31")
            self.sorted_tab = torch.zeros((int(be.c.gmx_sorted_uniforms_words(self.N)),), dtype=torch.int32, device=dev)
            hk = np.stack([self.step_keys[t][1].host() for t in range(T)]).astype(np.uint32)
            self.sorted_keys = torch.from_numpy(hk.view(np.int32)).to(dev)
        if self.comm and self.world > 1 and hasattr(self.dist, "barrier"):
            # ranks leave prepare() together (hiprtc compiles are seconds apart between ranks): the bounded waits of
            # the peer-mapped exchanges only ever see the microseconds of skew a running sweep has
            self.dist.barrier()
        return self

    def _alloc_exchange(self):
        dev, n, W, C = _lib.get().device, self.n, self.world, self.capacity
        if getattr(self, "peer_mode", False):
            # what crosses ranks lands in the communicator's fine-grained landing block; the extended states (and their
            # tails, filled by this rank's own routing launch) are ordinary memory
            self.peer_leaves = self.D * (2 if self.rejuvenate is not None else 1)
            if self.peer_leaves > _lib.PEER_MAX_LEAVES:
                raise NotImplementedError("ShardedBootstrapSweep over the fused peer exchange: at most "
                                          f"{_lib.PEER_MAX_LEAVES} routed leaves (GMX_PEER_MAX_LEAVES)")
            self.peer_land, _ = self.cx.landing(n, C, self.peer_leaves)              # COLLECTIVE
            if not hasattr(self, "peer_tag"):
                self.peer_tag, self.peer_status = self.cx.step_words()
        # extended state, double-buffered: [ n local | W*C received ]; ancestors index into it
        # destinations of collectives come from the communicator (peer-mapped memory under GENMI_COMM=p2p)
        mk = (lambda shape, dt=torch.float32: self.cx.alloc(shape, dt)) if (self.cx is not None and not getattr(self, "peer_mode", False)) else \
            (lambda shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev))
        self.xrows = [mk((self.D, n + W * C)) for _ in range(2)]
        self.xext = [r[0] for r in self.xrows]                    # component 0 (THE state when it is a scalar)
        self.send = torch.zeros((self.D, W * C), dtype=torch.float32, device=dev)
        self.idx = torch.zeros((n,), dtype=torch.int32, device=dev)
        if self.rejuvenate is not None:
            # aext[t % 2][:n] = the MH-moved, resampled state step t is extended from; its tail receives the
            # remote copies of it when it travels as the second routed leaf of the NEXT resampling
            self.arows = [mk((self.D, n + W * C)) for _ in range(2)]
            self.send2 = torch.zeros((self.D, W * C), dtype=torch.float32, device=dev)
        self._bound = [None] * self.T

    def _asrc(self, tb, local=False):
        """the MH-moved state of buffer tb as the model sees it: [n + W*C] (or [.., D]); local=True: this rank's n rows"""
        rows = self.arows[tb][:, :self.n] if local else self.arows[tb]
        return rows[0] if not self.event else rows.t()

    def _src(self, tb):
        """what the step model sees as the previous state (before the gather): [n + W*C] or [n + W*C, D]"""
        return self.xrows[tb][0] if not self.event else self.xrows[tb].t()

    # ------------------------------------------------------------------
    def _bind_step(self, t):
        """Everything step t launches, bound once (all buffers are persistent): the site-program
        launch arguments plus the argument tuples of the resampling entry points."""
        be = _lib.get()
        n, g, W, C = self.n, self.rank, self.world, self.capacity
        k_prop, k_res, _ = self.step_keys[t]
        obs = ChoiceMap.empty().set(self.obs_addr, self.ys[t])
        cur = self.xext[t % 2]
        mh = None
        shard_in = None
        if self.fuse_sh and t >= 1:
            # the launch that gathers ROUTES step t - 1 first: that step's log-weights, statistics, states and resampling key
            pw = (t - 1) % 2
            prev_rows = [self.xrows[pw][d] for d in range(self.D)]
            if self.rejuvenate is not None and t - 1 >= 1:         # second routed leaf: what x_{t-1} was extended from
                prev_rows += [self.arows[pw][d] for d in range(self.D)]
            kh_ = self.step_keys[t - 1][1].host()
            p_prev = _lib.Peer()
            p_prev.land_d, p_prev.tag_base_d, p_prev.status_d = (self.peer_land.data_ptr(), self.peer_tag.data_ptr(),
                                                                 self.peer_status.data_ptr())
            p_prev.rank, p_prev.world, p_prev.step, p_prev.tiles = g, W, t - 1, (n + CDF_TILE - 1) // CDF_TILE
            p_prev.capacity, p_prev.leaves = C, len(prev_rows)
            shard_in = dict(lw=self.lw_pp[pw], stats_own=self.stats_own_pp[pw], plan=self.plan, total_out=self.totals[t - 1:t],
                            max_out=self.maxs[t - 1:t], status=self.sh_status, shift=self.shift, tag=1 + (t - 1) % _lib.ANC_TAG_MAX,
                            key=(int(kh_[0]), int(kh_[1])), peer=p_prev, state=prev_rows,
                            tail=[r_[n:] for r_ in prev_rows])
        if t == 0:
            prog = self.p_init
            leaves = prog.leaves((), obs, self._noise_leaves(t, prog)) if self.noise_ahead else prog.leaves((), obs)
        elif self.rejuvenate is None:
            prog = self.p_step
            a_ = (Gathered(self._src((t - 1) % 2), self.idx),) + tuple(self.step_extra(t))
            leaves = prog.leaves(a_, obs, self._noise_leaves(t, prog)) if self.noise_ahead else prog.leaves(a_, obs)
        elif self.chain_mh:
            # ONE program: the MH move on the resampled particles of step t - 1 (keys split(k_mh, N)[g*n + i]), then the
            # extension to step t from the moved state (keys split(k_prop, N)[g*n + i]: OP_KSPLITU of two launch values)
            ch = ChoiceMap.empty().set(self.obs_addr, self.ys[t - 1]).set(self.state_addr,
                                                                          Gathered(self._src((t - 1) % 2), self.idx))
            ex_t = tuple(self.step_extra(t))
            kw = k_prop.host()
            if t == 1:
                prog = self.p_mhvm_init
                leaves = prog.leaves((), ch, self.rejuvenate, ex_t, obs, (int(kw[0]), int(kw[1])),
                                     self._noise_leaves(t, prog) if self.noise_ahead else ())
            else:
                prog = self.p_mhvm_step
                leaves = prog.leaves((Gathered(self._asrc((t - 1) % 2), self.idx),) + tuple(self.step_extra(t - 1)), ch,
                                     self.rejuvenate, ex_t, obs, (int(kw[0]), int(kw[1])),
                                     self._noise_leaves(t, prog) if self.noise_ahead else ())
            k_prop = self.step_keys[t][2]          # the launch key of the chained program is the MOVE's
        else:
            # the MH move on the resampled particles of step t-1, keys split(k_mh, N)[g*n + i]
            ch = ChoiceMap.empty().set(self.obs_addr, self.ys[t - 1]).set(self.state_addr,
                                                                          Gathered(self._src((t - 1) % 2), self.idx))
            if t == 1:
                mprog, mleaves = self.p_mh_init, self.p_mh_init.leaves((), ch, self.rejuvenate)
            else:
                mprog = self.p_mh_step
                mleaves = mprog.leaves((Gathered(self._asrc((t - 1) % 2), self.idx),) + tuple(self.step_extra(t - 1)), ch,
                                       self.rejuvenate)
            mbufs = [None] * len(mprog.comp.outputs)
            mbufs[mprog.ro[1]] = self.arows[t % 2][:, :n]
            mbufs[mprog.ao[1]] = self.accept.reshape(1, n)
            mh = (mprog.comp, mprog.comp.bind(mleaves, (n,), lazy_split(self.step_keys[t][2], self.N),
                                             out_buffers=mbufs, index_offset=g * n, shard_in=shard_in), mleaves)
            shard_in = None                       # (the move routed: the extension below reads its output locally)
            prog = self.p_step
            leaves = prog.leaves((self._asrc(t % 2, local=True),) + tuple(self.step_extra(t)), obs)
        bufs = [None] * len(prog.comp.outputs)
        rows_t = self.xrows[t % 2]
        bufs[prog.ro[1]] = rows_t[:, :n]                 # [D, n] window of the [D, n + W*C] rows
        if self.chain_mh and t >= 1:
            bufs[prog.mo[1]] = self.arows[t % 2][:, :n]     # the moved state step t was extended from
            bufs[prog.ao[1]] = self.accept.reshape(1, n)
        w_ = t % 2 if self.fuse_sh else 0                # (one launch per step: two sets of log-weights / statistics)
        lw_t = self.lw_pp[w_]
        bufs[prog.wo[1]] = lw_t.reshape(1, n)
        # keys of the GLOBAL particle index: split(k_prop, N)[g*n + i]
        writes_stats = self.tiles_mode and prog.comp.writes_tile_stats()
        peer = None
        if self.peer_mode:
            peer = _lib.Peer()
            peer.land_d, peer.tag_base_d, peer.status_d = self.peer_land.data_ptr(), self.peer_tag.data_ptr(), self.peer_status.data_ptr()
            peer.rank, peer.world, peer.step, peer.tiles = g, W, t, (n + CDF_TILE - 1) // CDF_TILE
            peer.capacity = C
            peer.leaves = self.D * (2 if (self.rejuvenate is not None and t >= 1) else 1)
        if writes_stats:        # the workgroup maxima land in the statistics block (red_out plane 0), the sums beside them
            st_w = t % 2 if self.fuse_sh else 0
            vm = prog.comp.bind(leaves, (n,), lazy_split(k_prop, self.N),
                                red_out=self.tile_max_pp[st_w] if self.fuse_sh else self.tile_max, out_buffers=bufs,
                                index_offset=g * n,
                                tile_stats=((self.tile_agg_pp[st_w] if self.fuse_sh else self.tile_agg), self.shift),
                                peer=peer, shard_in=shard_in)
        else:
            vm = prog.comp.bind(leaves, (n,), lazy_split(k_prop, self.N), red_out=self.partials, out_buffers=bufs,
                                index_offset=g * n)
        kh = k_res.host()
        kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
        m = self.maxs[t:t + 1]
        tot = self.totals[t:t + 1]
        P = be.ptr
        rows = int(be.c.gmx_program_grid(prog.comp.handle, n))        # block maxima the site program writes
        pmax = self.partials[0, :rows]
        step2 = recv2 = None
        if self.rejuvenate is not None and t >= 1:       # second routed leaf (per component): what x_t was extended from
            cur_a = self.arows[t % 2]
            step2 = [((self.kind, kk) if self.kind != MULTINOMIAL_SORTED else (P(self.sorted_tab),)) +
                     (P(self.totals_all), P(self.plan), P(tot), P(self.cdf), g, W, n, C, P(cur_a[d]),
                      P(self.send2[d]), P(self.idx)) for d in range(self.D)]
            recv2 = [cur_a[d][n:] for d in range(self.D)]
        tiles = None
        if self.tiles_mode:
            mk = lambda row, snd: (self.kind, kk, P(self.totals_all), P(self.plan), P(tot), P(self.lw), P(self.stats_own),
                                   P(m), self.shift, g, W, n, C, P(row), P(snd), P(self.idx))
            mkf = lambda row, snd: (self.kind, kk, P(self.stats_all), P(self.plan), P(tot), P(self.lw), P(m), self.shift,
                                    g, W, n, C, P(row), P(snd), P(self.idx))
            fused_ok = W <= 64 and self.fused_req
            pk = None
            if peer is not None:
                from ctypes import c_void_p as _vp
                rows_l = [rows_t[d] for d in range(self.D)]
                if self.rejuvenate is not None and t >= 1:
                    rows_l += [self.arows[t % 2][d] for d in range(self.D)]
                L = len(rows_l)
                st_arr = (_vp * L)(*[r.data_ptr() for r in rows_l])
                tl_arr = (_vp * L)(*[r[n:].data_ptr() for r in rows_l])
                so_t = self.stats_own_pp[t % 2] if self.fuse_sh else self.stats_own
                pk = {"peer": peer, "put_stats": None if writes_stats else (P(so_t), peer, n),
                      "step": (self.kind, kk, P(so_t), peer, P(self.plan), P(tot), P(lw_t), P(m), self.shift, n,
                               st_arr, tl_arr, P(self.idx)), "keep": (rows_l, st_arr, tl_arr)}
            tiles = {"peer": pk,
                     "stats": None if writes_stats else (P(self.lw), n, self.shift, P(self.tile_max), P(self.tile_agg)),
                     "totals": (P(self.stats_all), W, n, P(self.totals_all), P(m)),
                     # the first routed leaf derives the totals / global max itself (one launch less in the chain);
                     # further leaves of the same step reuse them
                     "fused": mkf(rows_t[0], self.send[0]) if fused_ok else None,
                     "steps": [mk(rows_t[d], self.send[d]) for d in range(self.D)],
                     "step2": [mk(self.arows[t % 2][d], self.send2[d]) for d in range(self.D)]
                     if (self.rejuvenate is not None and t >= 1) else None}
        return {
            "tiles": tiles,
            "prog": prog.comp, "vm": vm, "mh": mh, "step2": step2, "recv2": recv2,
            "pmax": pmax, "keep": (kk, tot, leaves, m),
            # the CDF kernel reduces the (all-reduced) block maxima itself and records the max in maxs[t]
            "cdf": (P(self.lw), n, self.shift, P(pmax), rows, P(m), P(self.cdf), P(self.total_d), P(self.ws)),
            # one routed leaf per state component: the same plan, D launches + D all-to-alls
            "steps": [((self.kind, kk) if self.sorted_tab is None else (P(self.sorted_tab),)) +
                      (P(self.totals_all), P(self.plan), P(tot), P(self.cdf), g, W, n, C,
                       P(rows_t[d]), P(self.send[d]), P(self.idx)) for d in range(self.D)],
            "sorted": None if self.sorted_tab is None else (P(self.sorted_keys[t]), 1, self.N, P(self.sorted_tab), 0),
            "recvs": [rows_t[d][n:] for d in range(self.D)],
        }

    def _step(self, t):
        be = _lib.get()
        b = self._bound[t]
        if b is None:
            b = self._bound[t] = self._bind_step(t)
        c, st = be.c, be.stream()
        if b["mh"] is not None:
            b["mh"][0].launch(b["mh"][1])                               # MH move on the resampled particles
        b["prog"].launch(b["vm"])                                      # x_t, lw_t, block maxima
        if b["tiles"] is not None:
            tl = b["tiles"]
            if tl["stats"] is not None:
                be.check(c.gmx_tile_stats(*tl["stats"], st), "gmx_tile_stats")
            if tl["peer"] is not None:          # no collective launch: puts + granule waits inside the two kernels
                pk = tl["peer"]
                if pk["put_stats"] is not None:
                    be.check(c.gmx_peer_put_stats(*pk["put_stats"], st), "gmx_peer_put_stats")
                if self.fuse_sh and t + 1 < self.T:
                    return                      # ONE launch per step: step t is routed by the launch of step t + 1
                be.check(c.gmx_shard_step_peer(*pk["step"], st), "gmx_shard_step_peer")
                return
            if self.comm:
                self.cx.all_gather(self.stats_all, self.stats_own)          # 12 bytes per 1024 particles per rank
            else:
                self.stats_all.copy_(self.stats_own)
            more = self.D > 1 or tl["step2"] is not None
            if tl["fused"] is None or more:
                be.check(c.gmx_shard_totals(*tl["totals"], st), "gmx_shard_totals")    # global max + every rank's total
            for d in range(self.D):
                if d == 0 and tl["fused"] is not None and not more:
                    be.check(c.gmx_shard_step_fused(*tl["fused"], st), "gmx_shard_step_fused")
                else:
                    be.check(c.gmx_shard_step_tiles(*tl["steps"][d], st), "gmx_shard_step_tiles")
                if self.comm:
                    self.cx.all_to_all(b["recvs"][d], self.send[d])
            if tl["step2"] is not None:
                for d in range(self.D):
                    be.check(c.gmx_shard_step_tiles(*tl["step2"][d], st), "gmx_shard_step_tiles")
                    if self.comm:
                        self.cx.all_to_all(b["recv2"][d], self.send2[d])
            return
        if self.comm:
            self.cx.all_reduce_max(b["pmax"])                            # element-wise MAX of the block maxima (<= 4 KB)
        be.check(c.gmx_weight_cdf(*b["cdf"], st), "gmx_weight_cdf")    # global max + local integer CDF against it
        if self.comm:
            self.cx.all_gather(self.totals_all, self.total_d)            # 8 bytes per rank
        else:
            self.totals_all.copy_(self.total_d)
        shard_step = c.gmx_shard_step
        if b["sorted"] is not None:
            be.check(c.gmx_sorted_uniforms(*b["sorted"], st), "gmx_sorted_uniforms")     # the step's table of all N slots
            shard_step = c.gmx_shard_step_sorted
        for d in range(self.D):
            be.check(shard_step(*b["steps"][d], st), "gmx_shard_step")       # slot boundaries + routing
            if self.comm:
                self.cx.all_to_all(b["recvs"][d], self.send[d])          # block s of recv <- block `me` of rank s
        if b["step2"] is not None:
            for d in range(self.D):
                be.check(shard_step(*b["step2"][d], st), "gmx_shard_step")
                if self.comm:
                    self.cx.all_to_all(b["recv2"][d], self.send2[d])

    def kernel_timers(self):
        """The site-program launch of a mid-sweep step (no collectives): bench.py's roofline kernel."""
        t = max(1, self.T // 2)

        def vm():
            n = self.n
            obs = ChoiceMap.empty().set(self.obs_addr, self.ys[t])
            prog = self.p_step
            a_ = (Gathered(self._src((t - 1) % 2), self.idx),) + tuple(self.step_extra(t))
            leaves = prog.leaves(a_, obs, self._noise_leaves(t, prog)) if self.noise_ahead else prog.leaves(a_, obs)
            bufs = [None] * len(prog.comp.outputs)
            bufs[prog.ro[1]] = self.xrows[t % 2][:, :n]
            bufs[prog.wo[1]] = self.lw.reshape(1, n)
            prog.comp.run(leaves, (n,), lazy_split(self.step_keys[t][0], self.N), red_out=self.partials,
                          out_buffers=bufs, index_offset=self.rank * n)
        return {"k_vm": vm}

    def enqueue(self):
        self._finished = False
        self.plan.zero_()
        if self.peer_mode:       # this sweep's tags: T more than the last one's (inside a captured graph too)
            be = _lib.get()
            be.check(be.c.gmx_peer_bump(be.ptr(self.peer_tag), self.T, be.stream()), "gmx_peer_bump")
        if self.noise_ahead:
            self._enqueue_noise_ahead()
        else:
            for t in range(self.T):
                self._step(t)
        self._enqueue_verdict()

    def _status_words(self):
        """device words whose non-zero value fails the sweep: the communicator's error word, the peer exchange's, the
        one-launch step's (an ancestor word that never arrived)"""
        words = []
        if self.comm and hasattr(self.cx, "failed_word"):
            words.append(self.cx.failed_word())
        if self.comm and self.peer_mode:
            words.append(self.peer_status[0:1])
        if self.fuse_sh:
            words.append(self.sh_status[0:1])
        return words

    def _enqueue_verdict(self):
        """the last node of a sweep: overflow flag and status words folded into ONE device word (gmx_sweep_verdict), so
        that finish() costs one read by the host (it was four reads and five small launches: 250 us of a 1.6-ms sweep)"""
        from ctypes import c_void_p
        be = _lib.get()
        words = self._status_words()
        arr = (c_void_p * max(1, len(words)))(*[c_void_p(w.data_ptr()) for w in words])
        be.check(be.c.gmx_sweep_verdict(be.ptr(self.plan[2:3]), arr, len(words), be.ptr(self.verdict), be.stream()),
                 "gmx_sweep_verdict")

    def capture(self):
        """OPT-IN across GPUs (`bench.py --rccl-graph`; the default at world size 1): capture the whole sweep — kernels AND the
        RCCL collectives, which comm.RcclComm issues on this same stream — into one hipGraph so
        the host leaves the loop.  Exercised at world size 1 only (no multi-GPU box in the build
        loop); the default multi-GPU path enqueues eagerly."""
        be = _lib.get()
        if self.comm and not self.cx.graph_safe:
            raise RuntimeError("capture() needs the direct RCCL communicator (torch.distributed's watchdog "
                               "queries events recorded inside the capture)")
        from ctypes import c_void_p
        s = torch.cuda.Stream(device=be.device)
        s.wait_stream(torch.cuda.current_stream(be.device))
        with torch.cuda.stream(s):
            self.enqueue()
            s.synchronize()
            be.check(be.c.gmx_capture_begin(be.stream()), "gmx_capture_begin")
            try:
                self.enqueue()
            finally:
                h = c_void_p()
                rc = be.c.gmx_capture_end(be.stream(), h)
            be.check(rc, "gmx_capture_end")
        torch.cuda.current_stream(be.device).wait_stream(s)
        self.graph = h
        return self

    def launch(self):
        if self.graph is not None:
            be = _lib.get()
            self._finished = False
            be.check(be.c.gmx_graph_launch(self.graph, be.stream()), "gmx_graph_launch")
        else:
            self.enqueue()

    def finish(self):
        """Read the overflow flag (one sync per sweep; COLLECTIVE — every rank calls it, which
        launch-then-state()/log_ml() on all ranks does); re-run with full capacity if it is set anywhere."""
        if self._finished:
            return self
        # a wait that timed out (a peer that never arrived) fails the sweep on EVERY rank instead of handing back stale
        # particles: the status words travel with the overflow flag — folded into `verdict` by the sweep's last node
        if self.comm:
            self.cx.all_reduce_max(self.verdict)
        flag = int(self.verdict.item())
        if flag >= 2:
            raise RuntimeError("ShardedBootstrapSweep: a peer's data did not arrive in time on some rank (the "
                               "peer-mapped exchange gave up waiting): the sweep's results are not valid")
        if flag != 0:
            self.reruns += 1
            self.capacity = self.n
            if self.graph is not None:
                # the captured sweep holds the old exchange buffers and the communicator's kernels: release it
                # (after the stream has drained) before the buffers go away — a dropped-but-live graph is what
                # made ncclCommDestroy hang at teardown; the re-run and later launches are eager
                be = _lib.get()
                if be.uses_streams:
                    torch.cuda.synchronize()
                be.c.gmx_graph_destroy(self.graph)
                self.graph = None
            self._alloc_exchange()
            self.enqueue()
            assert int(self.plan[2].item()) == 0
        self._finished = True
        return self

    def close(self):
        """Release the captured graph, then the direct RCCL communicator — in that order: a captured sweep
        holds the communicator's kernels and ncclCommDestroy waits on it forever otherwise (measured: a hang at
        teardown).  COLLECTIVE when a communicator exists."""
        be = _lib.get()
        if self.graph is not None:
            be.c.gmx_graph_destroy(self.graph)
            self.graph = None
        if self.cx is not None and hasattr(self.cx, "destroy") and getattr(self, "_own_cx", True):
            if be.uses_streams:
                torch.cuda.synchronize()
            self.cx.destroy()
        self.cx = None

    def log_ml(self) -> float:
        self.finish()
        acc = 0.0
        for m, tot in zip(self.maxs.cpu().tolist(), self.totals.cpu().numpy().view(np.uint64).tolist()):
            if tot == 0:
                return -math.inf
            acc += cdf_reference(m) + math.log(tot) - self.shift * math.log(2.0) - math.log(self.N)
        return acc

    def state(self):
        """this rank's resampled particles after the last step (global slots [g*n, (g+1)*n))"""
        self.finish()
        from ..engine import gather_leaves
        rows = self.xrows[(self.T - 1) % 2]
        got = gather_leaves([rows[d] for d in range(self.D)], self.idx)
        return got[0] if not self.event else torch.stack(got, dim=1)


class CountingComm:
    """Wraps a communicator and counts its collectives (tests; `sharded_importance_resample(..., stats=...)`)."""

    def __init__(self, inner):
        self.inner, self.counts = inner, {"all_reduce_max": 0, "all_gather": 0, "all_to_all": 0}
        self.rank, self.world, self.name = inner.rank, inner.world, inner.name

    def all_reduce_max(self, t):
        self.counts["all_reduce_max"] += 1
        return self.inner.all_reduce_max(t)

    def all_gather(self, out, inp):
        self.counts["all_gather"] += 1
        return self.inner.all_gather(out, inp)

    def all_to_all(self, out, inp):
        self.counts["all_to_all"] += 1
        return self.inner.all_to_all(out, inp)

    def alloc(self, shape, dtype=torch.float32):
        return self.inner.alloc(shape, dtype)


def sharded_importance_resample(target, k_per_rank: int, key: Key, dist, kind="systematic", capacity=None, comm=None,
                                stats: dict | None = None, cdf_form=False):
    """BASELINE config 4 across ranks: `ImportanceK(target, k_particles = world * k_per_rank).run_smc(key)`
    with rank g holding particles [g*k, (g+1)*k) (same key tree: keys split(sub, K)[g*k + i], so the
    ensemble is the single-process one), then ONE global resampling with ONE plan for the whole trace:

      1. tile statistics of the local log-weights (gmx_tile_stats)         -> all-gather (12 B per 1024 particles)
      2. gmx_shard_totals: global max + every rank's integer total; gmx_shard_step_tiles routes ONE 4-byte leaf:
         each particle's LOCAL INDEX — so `send_idx[d*C + k]` names the local particle that fills slot k of the block
         this rank ships to rank d, and `next_idx` says where each of this rank's slots finds its ancestor;
      3. every 4-byte row of every trace leaf (10 latents for 8-schools) is packed by `send_idx` into one
         [world, rows, C] buffer                                              -> ONE equal-split all-to-all
      4. one gather by `next_idx` over [local | received] rebuilds every leaf.

    Two data-path collectives per resampling whatever the number of leaves (the first form did one all-to-all and one
    routing launch PER ROW, plus a max all-reduce and a totals all-gather), and one 8-byte all-reduce for the
    capacity-overflow flag.  (More than 64 ranks or multinomial_sorted: the CDF-array form computes the
    same plan — all-reduce of the max, gmx_weight_cdf, all-gather of the totals, gmx_shard_step.)
    `stats`, if given, receives {"collectives": {...}, "rows": R, "capacity": C, "form": ...}.
    Returns (ParticleCollection of this rank's k resampled particles, this rank's pre-resampling log-weights).
    COLLECTIVE: every rank calls it."""
    _check_shard_alignment(int(k_per_rank), dist.get_world_size())
    from .smc import _KINDS, FUSED_RESAMPLE_MAX, LogMLOffset, ParticleCollection, trace_map
    from ..engine import materialize
    be = _lib.get()
    dev = be.device
    g, W = dist.get_rank(), dist.get_world_size()
    n, K = int(k_per_rank), int(k_per_rank) * dist.get_world_size()
    kind = _KINDS[kind] if isinstance(kind, str) else int(kind)
    if kind not in (SYSTEMATIC, STRATIFIED, MULTINOMIAL_SORTED):
        raise NotImplementedError("sharded_importance_resample: the router takes the ordered schemes (systematic, "
                                  "stratified, multinomial_sorted)")
    if comm is None and W > 1:
        from .comm import make_comm
        comm = make_comm(dist, dev)
    if comm is not None:
        comm = CountingComm(comm)
    key, sub = split(key)                                            # smc.py:299
    trs, lw = target.importance(lazy_split(sub, n, offset=g * n), ChoiceMap.empty())
    lw = lw.float().contiguous()
    shift = cdf_shift(K)
    kh = key.host()                                                  # resampling key: the algorithm's leftover `key`
    kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
    mx = torch.empty((1,), dtype=torch.float32, device=dev)
    calloc = (lambda shape, dt: comm.alloc(shape, dt)) if (comm is not None and hasattr(comm.inner, "alloc")) else \
        (lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev))          # destinations of collectives
    totals_all = calloc((W,), torch.int64)
    gtotal = torch.zeros((1,), dtype=torch.int64, device=dev)
    # (the tile-statistics plan takes any per-rank size: gmx_shard_totals / gmx_shard_step_tiles stride over the table)
    tiles_form = kind in (0, 1) and W <= 64 and not cdf_form
    if tiles_form:
        nbytes = int(be.c.gmx_shard_stats_bytes(n))
        tiles = (n + 1023) // 1024
        pad = tiles + (tiles & 1)
        stats_own = torch.zeros((nbytes,), dtype=torch.uint8, device=dev)
        stats_all = calloc((W * nbytes,), torch.uint8)
        be.check(be.c.gmx_tile_stats(be.ptr(lw), n, shift, be.ptr(stats_own[pad * 8:]), be.ptr(stats_own), be.stream()),
                 "gmx_tile_stats")
        if W > 1:
            comm.all_gather(stats_all, stats_own)
        else:
            stats_all.copy_(stats_own)
        be.check(be.c.gmx_shard_totals(be.ptr(stats_all), W, n, be.ptr(totals_all), be.ptr(mx), be.stream()),
                 "gmx_shard_totals")
        cdf = None
    else:
        # global max (deterministic LSE kernel's max output), local CDF against it, totals
        dummy = torch.empty((1,), dtype=torch.float32, device=dev)
        rows_ws = torch.empty(((be.c.gmx_logsumexp_workspace(1, n) + 3) // 4,), dtype=torch.int32, device=dev)
        be.check(be.c.gmx_logsumexp(be.ptr(lw), 1, n, be.ptr(dummy), be.ptr(mx), be.ptr(rows_ws), be.stream()), "gmx_logsumexp")
        if W > 1:
            comm.all_reduce_max(mx)
        cdf = torch.empty((n,), dtype=torch.int64, device=dev)
        total = torch.zeros((1,), dtype=torch.int64, device=dev)
        ws = torch.zeros(((be.c.gmx_weight_cdf_workspace(n) + 7) // 8,), dtype=torch.int64, device=dev)
        be.check(be.c.gmx_weight_cdf(be.ptr(lw), n, shift, None, 0, be.ptr(mx), be.ptr(cdf), be.ptr(total), be.ptr(ws),
                                     be.stream()), "gmx_weight_cdf")
        if W > 1:
            comm.all_gather(totals_all, total)
        else:
            totals_all.copy_(total)

    lazy_out = W == 1
    # ---- every per-particle leaf of the trace as 4-byte rows [R, n] ----
    # (ONE rank: nothing is shipped — every slot's ancestor is local — so no leaf is packed at all: the new trace's leaves
    #  are lazy gathers through next_idx, materialised when read, as smc.resample's are)
    specs, rows = [], []

    def collect(v):
        v = materialize(v)
        if tuple(v.shape[:1]) != (n,):
            specs.append(None)
            return v
        orig = v.dtype
        if v.element_size() == 8:                                     # int64 / float64: two 4-byte rows per value
            flat = v.reshape(n, -1).contiguous().view(torch.int32)
        elif v.element_size() == 4:
            flat = v.reshape(n, -1)
        elif orig in (torch.bool, torch.uint8, torch.int8, torch.int16):
            flat = v.to(torch.int32).reshape(n, -1)                   # small integers travel as i32 and come back as they were
        else:
            raise TypeError(f"sharded_importance_resample: cannot route a leaf of dtype {orig}")
        specs.append((orig, tuple(v.shape), flat.dtype, len(rows), flat.shape[1]))
        for c in range(flat.shape[1]):
            rows.append(flat[:, c].contiguous().view(torch.float32))
        return v
    if not lazy_out:
        trace_map(trs, collect)
    R = len(rows)
    table = torch.stack(rows) if R else torch.zeros((0, n), dtype=torch.float32, device=dev)      # [R, n]

    # (ONE rank ships nothing: no send blocks to size, no overflow to ask the device about — and no host sync)
    C = 1 if W == 1 else max(1, min(int(capacity) if capacity else max(4096, n // 32), n))
    sorted_tab = None
    while True:
        plan = torch.zeros((int(be.c.gmx_shard_plan_words(W)),), dtype=torch.int64, device=dev)
        next_idx = torch.zeros((n,), dtype=torch.int32, device=dev)
        send_idx = torch.zeros((W * C,), dtype=torch.int32, device=dev)              # unused capacity: particle 0 (valid)
        local_index = torch.arange(n, dtype=torch.int32, device=dev)                  # the ONE routed leaf
        if tiles_form:
            be.check(be.c.gmx_shard_step_tiles(kind, kk, be.ptr(totals_all), be.ptr(plan), be.ptr(gtotal), be.ptr(lw),
                                               be.ptr(stats_own), be.ptr(mx), shift, g, W, n, C, be.ptr(local_index),
                                               be.ptr(send_idx), be.ptr(next_idx), be.stream()), "gmx_shard_step_tiles")
        elif kind == MULTINOMIAL_SORTED:      # every rank draws the same table of the K global slots from the key
            if sorted_tab is None:
                sorted_tab = torch.zeros((int(be.c.gmx_sorted_uniforms_words(K)),), dtype=torch.int32, device=dev)
                kd = torch.tensor([int(kk[0]), int(kk[1])], dtype=torch.int64).to(torch.int32).to(dev)
                be.check(be.c.gmx_sorted_uniforms(be.ptr(kd), 1, K, be.ptr(sorted_tab), 0, be.stream()), "gmx_sorted_uniforms")
            be.check(be.c.gmx_shard_step_sorted(be.ptr(sorted_tab), be.ptr(totals_all), be.ptr(plan), be.ptr(gtotal),
                                                be.ptr(cdf), g, W, n, C, be.ptr(local_index), be.ptr(send_idx),
                                                be.ptr(next_idx), be.stream()), "gmx_shard_step_sorted")
        else:
            be.check(be.c.gmx_shard_step(kind, kk, be.ptr(totals_all), be.ptr(plan), be.ptr(gtotal), be.ptr(cdf), g, W,
                                         n, C, be.ptr(local_index), be.ptr(send_idx), be.ptr(next_idx), be.stream()),
                     "gmx_shard_step")
        if W == 1:
            break
        flag = plan[2:3].clone()
        comm.all_reduce_max(flag)
        if int(flag.item()) == 0:
            break
        C = n                                                        # always sufficient
    # ---- pack by destination, ONE all-to-all, one gather ----
    if W > 1 and R:
        packed = table[:, send_idx.long()].reshape(R, W, C).permute(1, 0, 2).contiguous()          # [W, R, C]
        recv = calloc(tuple(packed.shape), packed.dtype)
        comm.all_to_all(recv.view(-1), packed.view(-1))
        # the local ancestors by ONE gather of the table; the few slots whose ancestor arrived from another rank are then
        # filled from the received blocks (no [R, n + W * C] copy of the whole table in between)
        far = next_idx >= n
        moved = table[:, torch.where(far, torch.zeros_like(next_idx), next_idx).long()]              # [R, n]
        rp = far.nonzero().squeeze(1)
        if rp.numel():
            moved[:, rp] = recv.permute(1, 0, 2).reshape(R, W * C)[:, (next_idx[rp] - n).long()]
    else:
        moved = table[:, next_idx.long()] if R else table                                             # [R, n]
    it = iter(specs)

    def rebuild(v):
        sp = next(it)
        if sp is None:
            return v
        orig, shape, fdt, r0, cols = sp
        out = moved[r0:r0 + cols].t().contiguous().view(fdt)
        if torch.empty((), dtype=orig).element_size() == 8:
            out = out.contiguous().view(orig)
        elif out.dtype != orig:
            out = (out != 0) if orig == torch.bool else out.to(orig)
        return out.reshape(shape)
    if lazy_out:
        from ..engine import Gathered

        def lazy(v):
            v = materialize(v)
            return Gathered(v, next_idx) if isinstance(v, torch.Tensor) and tuple(v.shape[:1]) == (n,) else v
        new = trace_map(trs, lazy)
    else:
        new = trace_map(trs, lambda v: rebuild(materialize(v)))
    if stats is not None:
        stats.update(collectives=dict(comm.counts) if comm is not None else {}, rows=R, capacity=C,
                     form="tile statistics" if tiles_form else "cdf array")
    off = LogMLOffset().plus(mx, gtotal, shift, K)
    out = ParticleCollection(new, None, True, off, n_zero=n)
    return out, lw
