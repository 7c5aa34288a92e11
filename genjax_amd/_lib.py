"""ctypes binding of libgenmi_hip.so (include/genmi.h).

The product has exactly one compute backend: the HIP library for gfx950.  If
it is missing, or no GPU is visible, every entry point raises — there is no
CPU fallback.  (tests/hostsim installs its own object through `install()` to
exercise the host logic on machines without a GPU; nothing in this package
refers to it.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_uint32, c_uint64, c_void_p)

import torch

GMX_MAX_IN = 64
GMX_MAX_OUT = 64
GMX_MAX_TAB = 8
GMX_MAX_UNI = 64

ABI_VERSION = 8
ANC_TAG_SHIFT = 24          # a tagged ancestor word = {tag: bits 24..31 | index: bits 0..23} (csrc/gmx_offspring.h)
ANC_INDEX_MASK = (1 << ANC_TAG_SHIFT) - 1
ANC_TAG_MAX = 255
KEY_NONE, KEY_ARRAY, KEY_SPLIT, KEY_ROWSPLIT, KEY_BCAST = 0, 1, 2, 3, 4
RESAMPLE_SYSTEMATIC, RESAMPLE_STRATIFIED, RESAMPLE_MULTINOMIAL, RESAMPLE_MULTINOMIAL_TILED = 0, 1, 2, 3
RESAMPLE_MULTINOMIAL_SORTED = 4


class ResampleIn(Structure):
    """struct gmx_resample_in: the previous step's resampling folded into a gathering site program's launch"""
    _fields_ = [
        ("lw_d", c_void_p),
        ("tile_max_d", c_void_p),
        ("tile_agg_d", c_void_p),
        ("max_out_d", c_void_p),
        ("total_out_d", c_void_p),
        ("status_d", c_void_p),
        ("shift", c_int32),
        ("tag", c_uint32),
        ("key0", c_uint32),
        ("key1", c_uint32),
        ("u0", c_uint32),
        ("reserved_", c_uint32),
    ]


class Peer(Structure):
    """struct gmx_peer: the peers of a sharded SMC step (include/genmi.h "Fused peer exchange")"""
    _fields_ = [
        ("land_d", c_void_p),
        ("tag_base_d", c_void_p),
        ("status_d", c_void_p),
        ("rank", c_int32),
        ("world", c_int32),
        ("step", c_int32),
        ("tiles", c_int32),
        ("capacity", c_int64),
        ("leaves", c_int32),
        ("reserved_", c_int32),
    ]


PEER_MAX_LEAVES = 32         # GMX_PEER_MAX_LEAVES


class ShardIn(Structure):
    """struct gmx_shard_in: the routing of the previous step of a sharded sweep folded into a gathering site program"""
    _fields_ = [
        ("lw_d", c_void_p),
        ("stats_own_d", c_void_p),
        ("plan_d", c_void_p),
        ("total_out_d", c_void_p),
        ("max_out_d", c_void_p),
        ("status_d", c_void_p),
        ("shift", c_int32),
        ("tag", c_uint32),
        ("key0", c_uint32),
        ("key1", c_uint32),
        ("u0", c_uint32),
        ("reserved_", c_int32),
        ("peer", Peer),
        ("state_d", c_void_p * PEER_MAX_LEAVES),
        ("tail_d", c_void_p * PEER_MAX_LEAVES),
    ]


class RunArgs(Structure):
    """struct gmx_run_args"""
    _fields_ = [
        ("in_d", c_void_p * GMX_MAX_IN),
        ("out_d", c_void_p * GMX_MAX_OUT),
        ("tab_d", c_void_p * GMX_MAX_TAB),
        ("uni", c_uint32 * GMX_MAX_UNI),
        ("ancestors_d", c_void_p),
        ("key_mode", c_int32),
        ("key0", c_uint32),
        ("key1", c_uint32),
        ("keys_d", c_void_p),
        ("key_inner", c_int64),
        ("index_offset", c_int64),
        ("red_out_d", c_void_p),
        ("tile_agg_d", c_void_p),
        ("tile_shift", c_int32),
        ("reserved_", c_int32),
        ("step_stride", c_int64),
        ("rs", ResampleIn),
        ("sh", ShardIn),
        ("peer", Peer),
    ]


class GenmiError(RuntimeError):
    pass


_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
LIB_PATH = os.environ.get("GENMI_LIB") or os.path.join(_LIB_DIR, "libgenmi_hip.so")   # GENMI_LIB: tuning builds


class Backend:
    """Loaded C-ABI + the torch device its pointers live on."""

    def __init__(self, cdll, device: torch.device, uses_streams: bool):
        self.c = cdll
        self.device = device
        self.uses_streams = uses_streams
        self._proto()

    def _proto(self):
        c = self.c
        # the launch-argument struct must have ONE layout on both sides: a stale binding would hand the kernels shifted pointers
        c.gmx_run_args_bytes.restype = c_size_t
        if int(c.gmx_run_args_bytes()) != ctypes.sizeof(RunArgs):
            raise GenmiError(f"gmx_run_args is {c.gmx_run_args_bytes()} bytes in the library and {ctypes.sizeof(RunArgs)} in "
                             "this binding (genjax_amd/_lib.py): rebuild both from one include/genmi.h")
        c.gmx_version.restype = c_int
        c.gmx_last_error.restype = c_char_p
        c.gmx_threefry2x32_host.argtypes = [c_uint32, c_uint32, c_uint32, c_uint32, POINTER(c_uint32)]
        c.gmx_threefry2x32_host.restype = None
        c.gmx_split.argtypes = [POINTER(c_uint32), c_int64, c_int64, c_void_p, c_void_p]
        c.gmx_split_rows.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p]
        c.gmx_fold_in.argtypes = [c_void_p, c_uint32, c_int64, c_void_p, c_void_p]
        c.gmx_random_bits.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p]
        c.gmx_program_create.argtypes = [POINTER(c_uint32), c_size_t, POINTER(c_void_p)]
        c.gmx_program_destroy.argtypes = [c_void_p]
        c.gmx_program_specialize.argtypes = [c_void_p]
        c.gmx_program_is_specialized.argtypes = [c_void_p]
        c.gmx_program_code_hash.argtypes = [c_void_p]
        c.gmx_program_code_hash.restype = c_uint64
        c.gmx_program_despecialize.argtypes = [c_void_p, ctypes.c_char_p]
        c.gmx_jit_rejected_count.argtypes = []
        c.gmx_jit_rejected_count.restype = c_int64
        c.gmx_program_grid.argtypes = [c_void_p, c_int64]
        c.gmx_program_grid.restype = c_int64
        c.gmx_program_run.argtypes = [c_void_p, c_int64, POINTER(RunArgs), c_void_p]
        c.gmx_program_writes_tile_stats.argtypes = [c_void_p]
        c.gmx_program_set_background.argtypes = [c_void_p, c_uint32]
        c.gmx_program_set_fuse_resample.argtypes = [c_void_p]
        c.gmx_program_set_fuse_resample_loop.argtypes = [c_void_p]
        c.gmx_program_fuses_resample.argtypes = [c_void_p]
        c.gmx_program_set_fuse_shard_step.argtypes = [c_void_p]
        c.gmx_program_fuses_shard_step.argtypes = [c_void_p]
        c.gmx_program_resident_particles.argtypes = [c_void_p]
        c.gmx_program_resident_particles.restype = c_int64
        c.gmx_logsumexp_workspace.argtypes = [c_int64, c_int64]
        c.gmx_logsumexp_workspace.restype = c_size_t
        c.gmx_logsumexp.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_sum_rows_inorder.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]
        c.gmx_sum_rows_workspace.argtypes = [c_int64, c_int64]
        c.gmx_sum_rows_workspace.restype = c_size_t
        c.gmx_sum_rows.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
        c.gmx_reduce_max.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
        c.gmx_weight_cdf_workspace.argtypes = [c_int64]
        c.gmx_weight_cdf_workspace.restype = c_size_t
        c.gmx_weight_cdf.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p]
        c.gmx_ancestors.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_int64, c_uint64, c_void_p,
                                    c_int64, c_int64, c_int64, c_void_p, c_void_p]
        c.gmx_multinomial_workspace.argtypes = [c_int64]
        c.gmx_multinomial_workspace.restype = c_size_t
        c.gmx_multinomial.argtypes = [POINTER(c_uint32), c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
        c.gmx_resample_workspace.argtypes = [c_int64]
        c.gmx_resample_workspace.restype = c_size_t
        c.gmx_resample.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_tile_stats.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]
        c.gmx_resample_tiles.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]
        c.gmx_multinomial_tiled_workspace.argtypes = [c_int64]
        c.gmx_multinomial_tiled_workspace.restype = c_size_t
        c.gmx_multinomial_tiled.argtypes = [POINTER(c_uint32), c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]
        c.gmx_slot_uniforms.argtypes = [c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p]
        c.gmx_resample_tiles_u.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_sorted_uniforms_words.argtypes = [c_int64]
        c.gmx_sorted_uniforms_words.restype = c_size_t
        c.gmx_sorted_uniforms.argtypes = [c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p]
        c.gmx_resample_sorted.argtypes = [POINTER(c_uint32), c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_resample_sorted_p.argtypes = [POINTER(c_uint32), c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_tile_prefix_words.argtypes = [c_int64]
        c.gmx_tile_prefix_words.restype = c_size_t
        c.gmx_tile_prefix.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
        c.gmx_resample_tiles_p.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]
        c.gmx_shard_stats_bytes.argtypes = [c_int64]
        c.gmx_shard_stats_bytes.restype = c_size_t
        c.gmx_shard_totals.argtypes = [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]
        c.gmx_shard_step_tiles.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                           c_void_p]
        c.gmx_shard_step_fused.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_shard_plan_words.argtypes = [c_int]
        c.gmx_shard_plan_words.restype = c_size_t
        c.gmx_shard_plan.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                                     c_void_p]
        c.gmx_shard_route.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_void_p, c_int, c_int, c_int64, c_int64,
                                      c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_shard_step.argtypes = [c_int, POINTER(c_uint32), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_shard_step_sorted.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64,
                                            c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        c.gmx_peer_landing_bytes.argtypes = [c_int, c_int64, c_int64, c_int]
        c.gmx_peer_landing_bytes.restype = c_size_t
        c.gmx_peer_bump.argtypes = [c_void_p, c_int32, c_void_p]
        c.gmx_sweep_verdict.argtypes = [c_void_p, POINTER(c_void_p), c_int32, c_void_p, c_void_p]
        c.gmx_peer_put_stats.argtypes = [c_void_p, Peer, c_int64, c_void_p]
        c.gmx_shard_step_peer.argtypes = [c_int, POINTER(c_uint32), c_void_p, Peer, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int64, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p]
        c.gmx_p2p_alloc.argtypes = [c_size_t, POINTER(c_void_p), c_void_p]
        c.gmx_p2p_open.argtypes = [c_void_p, POINTER(c_void_p)]
        c.gmx_p2p_close.argtypes = [c_void_p]
        c.gmx_p2p_free.argtypes = [c_void_p]
        c.gmx_p2p_exchange.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_size_t, c_void_p]
        c.gmx_gather.argtypes = [POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), c_int32,
                                 c_void_p, c_int64, c_void_p]
        c.gmx_categorical_rows.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]
        c.gmx_mh_accept.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
        c.gmx_select.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                 POINTER(c_int32), c_int32, c_int64, c_void_p]
        c.gmx_capture_begin.argtypes = [c_void_p]
        c.gmx_capture_end.argtypes = [c_void_p, POINTER(c_void_p)]
        c.gmx_graph_launch.argtypes = [c_void_p, c_void_p]
        c.gmx_graph_destroy.argtypes = [c_void_p]
        c.gmx_timer_create.argtypes = [POINTER(c_void_p)]
        c.gmx_timer_start.argtypes = [c_void_p, c_void_p]
        c.gmx_timer_stop.argtypes = [c_void_p, c_void_p]
        c.gmx_timer_elapsed_ms.argtypes = [c_void_p, POINTER(c_float)]
        c.gmx_timer_destroy.argtypes = [c_void_p]

    # ------------------------------------------------------------------
    def check(self, rc: int, what: str = ""):
        if rc != 0:
            msg = self.c.gmx_last_error()
            raise GenmiError(f"{what}: {msg.decode() if msg else 'error'}")

    def stream(self):
        if not self.uses_streams:
            return None
        return c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def ptr(self, t):
        if t is None:
            return None
        if t.device.type != self.device.type:
            raise GenmiError(f"tensor on {t.device}, backend on {self.device}")
        return c_void_p(t.data_ptr())


_backend: Backend | None = None


def install(backend: Backend | None):
    """Replace the active backend (used by tests/hostsim only)."""
    global _backend
    _backend = backend


def get() -> Backend:
    global _backend
    if _backend is not None:
        return _backend
    if not os.path.exists(LIB_PATH):
        raise GenmiError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). genjax_amd has no CPU fallback.")
    if not torch.cuda.is_available():
        raise GenmiError("no HIP device visible: genjax_amd runs on MI355X (gfx950) only; "
                         "there is no CPU fallback.")
    cdll = ctypes.CDLL(LIB_PATH)
    dev = torch.device("cuda", torch.cuda.current_device())
    _backend = Backend(cdll, dev, uses_streams=True)
    if _backend.c.gmx_version() != ABI_VERSION:
        raise GenmiError("libgenmi_hip.so ABI version mismatch")
    return _backend


def device() -> torch.device:
    return get().device
