"""Particle ensembles sharded across the GPUs of one node.

One process per GPU (`torch.distributed`; backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Rank g owns the contiguous block of global
particle indices [g*n, (g+1)*n).  Nothing about the result depends on the
number of ranks:

  * per-particle keys are derived from the GLOBAL index (GMX_KEY_SPLIT with
    index_offset = g*n);
  * weights are exact integers q_i = floor(exp(lw_i - M) * 2^shift) with the
    GLOBAL max M (all-reduce MAX of 4 bytes) and shift from the GLOBAL count;
  * the global CDF is offset_g + local CDF with offset_g from an all-gather of
    the 8-byte local totals (exact integer sums: any partition gives the same CDF);
  * rank r resolves the output slots that fall into ITS mass interval
    [offset_r, offset_r + total_r) — for systematic resampling a contiguous slot
    range [S_r, E_r) known on every rank from the totals alone — gathers its own
    states for them and ships them to the slot owners with ONE all-to-all-v of
    states (the only bulk exchange; balanced weights keep most of it rank-local).

So per SMC step: 1 all-reduce (4 B), 1 all-gather (8 B/rank), 1 all-to-all-v
(<= 4*D bytes per particle, mostly self-sends).  xGMI is point-to-point, so the
all-to-all-v maps onto direct peer links rather than a ring.
"""
from __future__ import annotations

import math
from ctypes import c_uint32

import numpy as np
import torch

from .. import _lib
from ..core.choice_map import ChoiceMap
from ..random import Key, fold_in, lazy_split, split
from .smc import SYSTEMATIC, cdf_shift


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


class ShardedBootstrapSweep:
    """smc.BootstrapSweep over `dist.get_world_size()` ranks, n particles per rank."""

    def __init__(self, init, step, n_per_rank: int, T: int, dist, obs_addr="y", step_extra=None, specialize=True):
        self.init, self.step, self.n, self.T, self.dist = init, step, int(n_per_rank), int(T), dist
        self.obs_addr = obs_addr
        self.step_extra = step_extra or (lambda t: ())
        self.specialize = specialize
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.N = self.n * self.world

    def prepare(self, key: Key, ys: torch.Tensor):
        from ..static import MinimalGenerate
        be = _lib.get()
        n, T, dev = self.n, self.T, be.device
        self.ys = ys.to(dev).float().contiguous()
        self.x = torch.zeros((n,), dtype=torch.float32, device=dev)          # this rank's current particles
        self.x_new = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.lw = torch.zeros((n,), dtype=torch.float32, device=dev)
        self.cdf = torch.zeros((n,), dtype=torch.int64, device=dev)
        self.max_d = torch.zeros((1,), dtype=torch.float32, device=dev)
        self.total_d = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.gtotal_d = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.totals_all = torch.zeros((self.world,), dtype=torch.int64, device=dev)
        self.ws = torch.zeros(((be.c.gmx_weight_cdf_workspace(n) + 7) // 8,), dtype=torch.int64, device=dev)
        self.shift = cdf_shift(self.N)
        obs0 = ChoiceMap.empty().set(self.obs_addr, self.ys[0])
        self.p_init = MinimalGenerate(self.init, (), obs0, (n,))
        self.p_step = MinimalGenerate(self.step, (self.x,) + tuple(self.step_extra(1)), obs0, (n,))
        if self.specialize and be.uses_streams:
            self.p_init.comp.specialize()
            self.p_step.comp.specialize()
        grid = be.c.gmx_program_grid(self.p_step.comp.handle, n)
        self.partials = torch.zeros((2, grid), dtype=torch.float32, device=dev)
        self.step_keys = []
        for t in range(T):
            ks = split(fold_in(key, t), 3)
            self.step_keys.append((ks[0], ks[1], ks[2]))
        self.maxs, self.totals = [], []
        return self

    # ------------------------------------------------------------------
    def _step(self, t):
        be, dist = _lib.get(), self.dist
        n, N, g, G = self.n, self.N, self.rank, self.world
        k_prop, k_res, _ = self.step_keys[t]
        obs = ChoiceMap.empty().set(self.obs_addr, self.ys[t])
        if t == 0:
            prog, leaves = self.p_init, self.p_init.leaves((), obs)
        else:
            prog = self.p_step
            leaves = prog.leaves((self.x,) + tuple(self.step_extra(t)), obs)
        bufs = [None] * len(prog.comp.outputs)
        bufs[prog.ro[1]] = self.x_new.reshape(1, n)
        bufs[prog.wo[1]] = self.lw.reshape(1, n)
        # keys of the GLOBAL particle index: split(k_prop, N)[g*n + i]
        prog.comp.run(leaves, (n,), lazy_split(k_prop, N), red_out=self.partials, out_buffers=bufs,
                      index_offset=g * n)
        # ---- global max: local reduce + all-reduce MAX (4 bytes) ----
        be.check(be.c.gmx_reduce_max(be.ptr(self.partials), self.partials.shape[1], be.ptr(self.max_d), be.stream()),
                 "gmx_reduce_max")
        dist.all_reduce(self.max_d, op=dist.ReduceOp.MAX)
        # ---- local integer CDF relative to the global max ----
        be.check(be.c.gmx_weight_cdf(be.ptr(self.lw), n, self.shift, None, 0, be.ptr(self.max_d), be.ptr(self.cdf),
                                     be.ptr(self.total_d), be.ptr(self.ws), be.stream()), "gmx_weight_cdf")
        # ---- all-gather the local totals (8 bytes per rank); offsets on every rank ----
        dist.all_gather_into_tensor(self.totals_all, self.total_d)
        tot = [int(v) & 0xFFFFFFFFFFFFFFFF for v in self.totals_all.cpu().tolist()]      # the one host sync
        offs = [0]
        for v in tot:
            offs.append(offs[-1] + v)
        total = offs[-1]
        self.maxs.append(float(self.max_d.item()))
        self.totals.append(total)
        # ---- which slots fall into which rank's mass (exact, from the totals alone) ----
        kh = k_res.host()
        from ..random import threefry2x32
        b0, b1 = threefry2x32(kh[0], kh[1], 0, 0)
        u0 = (int(b0) ^ int(b1)) >> 9
        bounds = systematic_slot_bounds(offs, total, N, u0)           # bounds[r] = f(offset_r); len G+1
        bounds[-1] = N
        S, E = bounds[g], bounds[g + 1]
        n_mine = E - S
        # ---- ancestors (local indices) of my slots, then my states for them ----
        self.gtotal_d.fill_(0)
        self.gtotal_d += torch.tensor([total if total < (1 << 63) else total - (1 << 64)], dtype=torch.int64,
                                      device=self.gtotal_d.device)
        send = torch.empty((max(n_mine, 1),), dtype=torch.float32, device=self.x.device)
        if n_mine > 0:
            anc = torch.empty((n_mine,), dtype=torch.int32, device=self.x.device)
            kk = (c_uint32 * 2)(int(kh[0]), int(kh[1]))
            be.check(be.c.gmx_ancestors(SYSTEMATIC, kk, be.ptr(self.cdf), n, offs[g] & 0xFFFFFFFFFFFFFFFF,
                                        be.ptr(self.gtotal_d), N, S, n_mine, be.ptr(anc), be.stream()),
                     "gmx_ancestors")
            from ..engine import gather_leaves
            send = gather_leaves([self.x_new], anc)[0].contiguous()
        # ---- all-to-all-v of states: slot owners are contiguous blocks of n ----
        in_splits, out_splits = [], []
        for r in range(G):
            lo, hi = max(S, r * n), min(E, (r + 1) * n)
            out_splits.append(max(0, hi - lo))                        # what I send to rank r
            lo, hi = max(bounds[r], g * n), min(bounds[r + 1], (g + 1) * n)
            in_splits.append(max(0, hi - lo))                         # what rank r sends to me
        assert sum(in_splits) == n, (in_splits, bounds)
        recv = torch.empty((n,), dtype=torch.float32, device=self.x.device)
        dist.all_to_all_single(recv, send[:n_mine] if n_mine > 0 else send[:0], output_split_sizes=in_splits,
                               input_split_sizes=out_splits)
        self.x = recv

    def launch(self):
        self.maxs, self.totals = [], []
        for t in range(self.T):
            self._step(t)

    def log_ml(self) -> float:
        acc = 0.0
        for m, tot in zip(self.maxs, self.totals):
            acc += m + math.log(tot) - self.shift * math.log(2.0) - math.log(self.N)
        return acc

    def state(self):
        """this rank's resampled particles after the last step (global slots [g*n, (g+1)*n))"""
        return self.x
