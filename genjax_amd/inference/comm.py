"""The three collectives a sharded SMC step needs, behind one small interface.

  all_reduce_max(t)      in place, 1 element (the global max log-weight; the overflow flag)
  all_gather(out, inp)   out[world * k] <- every rank's inp[k] (the integer totals; the tile statistics)
  all_to_all(out, inp)   equal split: block s of out <- block `me` of rank s's inp (states)

`RcclComm` calls RCCL directly (ctypes on the librccl.so torch already loaded) on
the CURRENT torch stream: ~5 us of host time per call instead of the 20-30 us a
`torch.distributed` call costs — with three collectives per 30-us SMC step the
host is the bottleneck, so this matters — and, because the calls sit on the same
stream as the kernels, the whole sweep is one ordered stream of work (capturable
into a hipGraph).  The communicator is bootstrapped through torch.distributed
(one broadcast of the 128-byte unique id), so launch stays `torch.distributed.run`,
one process per GPU.  `TorchComm` is the same interface over a torch.distributed
process group ("gloo" in the CPU tests; fallback on the GPU box).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char, c_int, c_size_t, c_void_p

import torch

_NCCL_INT8, _NCCL_INT64, _NCCL_FLOAT32, _NCCL_MAX = 0, 4, 7, 2


class TorchComm:
    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        pg = getattr(getattr(dist, "group", None), "WORLD", None)
        if pg is not None and hasattr(pg, "_allgather_base") and hasattr(pg, "alltoall_base"):
            # straight to the ProcessGroup: skips the module-level wrappers' per-call checks
            opts = dist.AllreduceOptions()
            opts.reduceOp = dist.ReduceOp.MAX
            self.all_reduce_max = lambda t: pg.allreduce([t], opts).wait()
            self.all_gather = lambda out, inp: pg._allgather_base(out, inp).wait()
            self.all_to_all = lambda out, inp: pg.alltoall_base(out, inp, [], []).wait()
        else:
            self.all_reduce_max = lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self.all_gather = lambda out, inp: dist.all_gather_into_tensor(out, inp)
            self.all_to_all = lambda out, inp: dist.all_to_all_single(out, inp)

    name = "torch.distributed"
    graph_safe = False


def _alloc_plain(self, shape, dtype=torch.float32):
    """destination buffers of collectives: ordinary device memory for the library communicators"""
    from .. import _lib
    return torch.zeros(tuple(shape), dtype=dtype, device=_lib.get().device)


TorchComm.alloc = _alloc_plain


class P2PComm:
    """GENMI_COMM=p2p — the collectives of a sharded SMC step as ONE launch each over peer-mapped memory
    (include/genmi.h "Peer-mapped exchange"): a collective copies this rank's blocks straight into its
    peers' LANDING buffers over xGMI (fine-grained device memory, IPC-mapped into every peer: two halves that alternate
    with the epoch's parity), raises a flag per peer, waits for its own flags and copies what landed into the caller's
    ordinary destination tensor.  No RCCL kernel, no host involvement, and — since the epoch lives on the device —
    capturable into the sweep's hipGraph.  Landing buffers are made on the first collective of each size (a COLLECTIVE
    step: the IPC handles are all-gathered; it happens in the eager warm-up pass, never inside a capture).
    Status: world size 1 runs on the device; world sizes 2 and 4 run through the tests' CPU mirror over process-shared
    memory (tests/test_distributed_cpu.py); across GPUs it is UNMEASURED (no multi-GPU box in the build loop)."""

    name = "p2p (peer-mapped xGMI, one launch per collective)"
    graph_safe = True
    alloc = _alloc_plain

    def __init__(self, dist, device: torch.device):
        from .. import _lib
        self.be = _lib.get()
        self.dist, self.device = dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self._allocs = []        # (base ptr, nbytes, [peer base ptrs], keep-alive, [opened peer mappings])
        self._landing = {}       # bytes per block -> (local landing tensor, device table of the peers' landing bases)
        self.flags, self._flag_table = self._shared((self.world,), torch.int64)
        self.state = torch.zeros((3 + self.world,), dtype=torch.int64, device=device)     # epoch, error, tickets

    # -- peer-mapped memory --------------------------------------------------------------------------
    def _shared(self, shape, dtype):
        """(a zeroed tensor in fine-grained, peer-mapped memory, the device table [world] of its address in every
        peer's allocation of the same call); COLLECTIVE"""
        n = 1
        for d in shape:
            n *= int(d)
        item = torch.empty((), dtype=dtype).element_size()
        nbytes = max(16, ((n * item + 15) // 16) * 16)
        ptr = c_void_p()
        handle = (ctypes.c_uint8 * 64)()
        self.be.check(self.be.c.gmx_p2p_alloc(nbytes, byref(ptr), handle), "gmx_p2p_alloc")
        peers = [ptr.value] * self.world
        opened = []
        if self.world > 1:
            mine = torch.tensor(list(handle), dtype=torch.uint8)
            gathered = [torch.zeros(64, dtype=torch.uint8) for _ in range(self.world)]
            if self.device.type == "cuda" and self.dist.get_backend() != "gloo":
                g_dev = [t.to(self.device) for t in gathered]
                self.dist.all_gather(g_dev, mine.to(self.device))
                gathered = [t.cpu() for t in g_dev]
            else:               # (a gloo group on a GPU box: the 64-byte handles travel as host tensors)
                self.dist.all_gather(gathered, mine)
            for s_, h in enumerate(gathered):
                if s_ == self.rank:
                    continue
                hb = (ctypes.c_uint8 * 64)(*h.tolist())
                p = c_void_p()
                self.be.check(self.be.c.gmx_p2p_open(hb, byref(p)), "gmx_p2p_open")
                peers[s_] = p.value
                opened.append(p.value)
        # the memory is only ever touched by the exchange kernel: a raw address is all that is needed (a tensor built
        # over it through __cuda_array_interface__ may be a COPY — measured: flags that never change, every wait timing out)
        table = torch.tensor(peers, dtype=torch.int64).to(self.device)
        self._allocs.append((ptr.value, nbytes, peers, None, opened))
        return c_void_p(ptr.value), table

    def _landing_for(self, nbytes: int):
        ent = self._landing.get(nbytes)
        if ent is None:
            if self.be.uses_streams and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("P2PComm: the first collective of a size allocates its landing buffers (a collective "
                                   "step): run the sweep once eagerly before capturing it")
            ent = self._landing[nbytes] = self._shared((2 * self.world * nbytes,), torch.uint8)
        return ent

    # -- collectives ------------------------------------------------------------------------------
    def _exchange(self, out, inp, stride_bytes, nbytes):
        be = self.be
        land, table = self._landing_for(nbytes)
        be.check(be.c.gmx_p2p_exchange(be.ptr(inp), stride_bytes, be.ptr(table), land, be.ptr(out),
                                       be.ptr(self._flag_table), self.flags, be.ptr(self.state), self.rank,
                                       self.world, nbytes, be.stream()), "gmx_p2p_exchange")

    def all_gather(self, out, inp):
        assert out.is_contiguous() and inp.is_contiguous()
        nbytes = inp.numel() * inp.element_size()
        assert out.numel() * out.element_size() == self.world * nbytes
        self._exchange(out, inp, 0, nbytes)

    def all_to_all(self, out, inp):
        assert out.is_contiguous() and inp.is_contiguous() and inp.numel() % self.world == 0
        nbytes = inp.numel() * inp.element_size() // self.world
        self._exchange(out, inp, nbytes, nbytes)

    def all_reduce_max(self, t):
        """the once-per-sweep overflow flag: off the hot path, through torch.distributed"""
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)

    def failed(self) -> bool:
        return bool(int(self.state[1].item()) != 0)

    def failed_word(self):
        """the error word itself (a device view): a sweep folds it into its one end-of-sweep read"""
        return self.state[1:2]

    def destroy(self):
        for _base, _n, _peers, _raw, opened in self._allocs:
            for p in opened:
                self.be.c.gmx_p2p_close(c_void_p(p))
        if self.world > 1:
            self.dist.barrier()                # nobody unmaps a segment a peer may still be writing
        for base, _n, _peers, _raw, _o in self._allocs:
            self.be.c.gmx_p2p_free(c_void_p(base))
        self._allocs, self._landing = [], {}


class PeerComm(P2PComm):
    """GENMI_COMM=peer — the FUSED peer exchange (include/genmi.h "Fused peer exchange"): a sharded SMC
    step has no collective launch at all.  The site program's epilogue puts its tile statistics into the other ranks'
    landing tables, gmx_shard_step_peer reads them as they arrive, puts offspring states into the owners' landing
    blocks and waits for the few values its own slots need — 8-byte tagged granules, no flags, no fences.  This class
    only owns the landing blocks (fine-grained, IPC-mapped), the tag word and the status word; one-off collectives
    (config 4's global resample, the overflow flag) go through the peer-mapped exchange it inherits."""

    name = "peer (fused puts: no collective launch per step)"
    fused = True

    def landing(self, n_per_rank: int, capacity: int, leaves: int):
        """(device table [world] of every rank's landing block as mapped here, bytes); COLLECTIVE"""
        nbytes = int(self.be.c.gmx_peer_landing_bytes(self.world, int(n_per_rank), int(capacity), int(leaves)))
        _own, table = self._shared((nbytes,), torch.uint8)
        return table, nbytes

    def step_words(self):
        """(tag base, status): two local device words of one sweep — tags start at 1 (zeroed granules never match)"""
        return (torch.ones((1,), dtype=torch.int32, device=self.device),
                torch.zeros((1,), dtype=torch.int64, device=self.device))


class _UniqueId(Structure):
    _fields_ = [("internal", c_char * 128)]


class RcclComm:
    """RCCL through its C API (rccl.h: ncclAllReduce :556, ncclAllGather :668, ncclAllToAll :790)."""

    name = "rccl (direct)"
    graph_safe = True

    def __init__(self, dist, device: torch.device):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        lib = ctypes.CDLL(path)
        lib.ncclGetUniqueId.argtypes = [POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [POINTER(c_void_p), c_int, _UniqueId, c_int]
        lib.ncclAllReduce.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]
        lib.ncclAllGather.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]
        lib.ncclAllToAll.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p]
        lib.ncclCommDestroy.argtypes = [c_void_p]
        lib.ncclGetErrorString.restype = ctypes.c_char_p
        self.lib, self.device = lib, device
        uid = _UniqueId()
        if self.rank == 0:
            self._check(lib.ncclGetUniqueId(byref(uid)), "ncclGetUniqueId")
        buf = torch.tensor(list(ctypes.string_at(byref(uid), 128)), dtype=torch.uint8).to(device)
        if self.world > 1:
            dist.broadcast(buf, src=0)
        ctypes.memmove(byref(uid), bytes(buf.cpu().tolist()), 128)
        comm = c_void_p()
        self._check(lib.ncclCommInitRank(byref(comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.comm = comm

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()}")

    def _stream(self):
        return c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def all_reduce_max(self, t):
        dt = _NCCL_FLOAT32 if t.dtype == torch.float32 else _NCCL_INT64
        assert t.dtype in (torch.float32, torch.int64) and t.is_contiguous()
        p = c_void_p(t.data_ptr())
        self._check(self.lib.ncclAllReduce(p, p, t.numel(), dt, _NCCL_MAX, self.comm, self._stream()), "ncclAllReduce")

    def all_gather(self, out, inp):
        """any dtype (gathered as bytes): the 8-byte totals, or a rank's block of tile statistics"""
        assert out.dtype == inp.dtype and out.numel() == self.world * inp.numel()
        assert out.is_contiguous() and inp.is_contiguous()
        self._check(self.lib.ncclAllGather(c_void_p(inp.data_ptr()), c_void_p(out.data_ptr()),
                                           inp.numel() * inp.element_size(), _NCCL_INT8, self.comm, self._stream()),
                    "ncclAllGather")

    def all_to_all(self, out, inp):
        assert out.dtype == inp.dtype == torch.float32 and out.numel() == inp.numel()
        assert inp.numel() % self.world == 0 and out.is_contiguous() and inp.is_contiguous()
        self._check(self.lib.ncclAllToAll(c_void_p(inp.data_ptr()), c_void_p(out.data_ptr()),
                                          inp.numel() // self.world, _NCCL_FLOAT32, self.comm, self._stream()),
                    "ncclAllToAll")

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None

    alloc = _alloc_plain


def _selftest(comm, device):
    """The three collectives on known patterns (a few bytes each): a communicator that answers wrongly is
    rejected here, where make_comm can still fall back to torch.distributed on every rank."""
    r, W = comm.rank, comm.world
    t = torch.tensor([float(r)], dtype=torch.float32, device=device)
    comm.all_reduce_max(t)
    g_in = torch.tensor([r], dtype=torch.int64, device=device)
    g_out = torch.zeros((W,), dtype=torch.int64, device=device)
    comm.all_gather(g_out, g_in)
    a_in = torch.tensor([r * 100 + d * 10 + j for d in range(W) for j in range(2)], dtype=torch.float32, device=device)
    a_out = torch.zeros_like(a_in)
    comm.all_to_all(a_out, a_in)
    want = torch.tensor([s_ * 100 + r * 10 + j for s_ in range(W) for j in range(2)], dtype=torch.float32)
    if float(t.item()) != float(W - 1) or g_out.cpu().tolist() != list(range(W)) or not torch.equal(a_out.cpu(), want):
        raise RuntimeError("communicator self-test failed")


class _Deadline:
    """Bounded time for a communicator's bootstrap + self-test: a collective that never completes (a peer that
    did not join, a transport that wedged) would otherwise hang the job until someone kills it.  When the
    deadline passes the process EXITS with a non-zero status (os._exit: no re-exec, no attempt to unwind a
    thread stuck inside a collective) after saying why on stderr; the launcher then tears the other ranks down."""

    def __init__(self, seconds: float, what: str):
        import threading
        self.what, self.seconds = what, seconds
        self._done = threading.Event()
        self._thread = threading.Thread(target=self._watch, daemon=True)

    def _watch(self):
        if not self._done.wait(self.seconds):
            import sys
            sys.stderr.write(f"genjax_amd: {self.what} did not finish within {self.seconds:.0f} s "
                             "(GENMI_COMM_TIMEOUT); exiting with status 3\n")
            sys.stderr.flush()
            os._exit(3)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False


def make_comm(dist, device: torch.device):
    """RCCL direct on a GPU box (GENMI_COMM=torch forces the torch.distributed path); every rank
    takes the same branch: the outcome of the RCCL bootstrap is agreed with a MIN all-reduce.
    Bootstrap and self-test run under a deadline (GENMI_COMM_TIMEOUT seconds, default 120; see _Deadline)."""
    want = os.environ.get("GENMI_COMM", "rccl" if device.type == "cuda" else "torch")
    timeout = float(os.environ.get("GENMI_COMM_TIMEOUT", "120"))
    if want in ("p2p", "peer"):
        with _Deadline(timeout, "the peer-mapped communicator bootstrap"):
            comm = (PeerComm if want == "peer" else P2PComm)(dist, device)
            _selftest(comm, device)
            if comm.failed():
                raise RuntimeError("the peer-mapped communicator's self-test timed out")
        return comm
    if want != "rccl" or device.type != "cuda":
        comm = TorchComm(dist)
        with _Deadline(timeout, "the torch.distributed communicator self-test"):
            _selftest(comm, device)
        return comm
    ok, comm = 1, None
    try:
        with _Deadline(timeout, "the direct RCCL communicator bootstrap / self-test"):
            comm = RcclComm(dist, device)
            _selftest(comm, device)
            torch.cuda.synchronize(device)
    except Exception as e:                       # missing symbol, bootstrap failure, a wrong answer, ...
        import warnings
        warnings.warn(f"direct RCCL communicator unavailable ({e!r}); using torch.distributed")
        ok = 0
    if dist.get_world_size() > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    return comm if ok and comm is not None else TorchComm(dist)
