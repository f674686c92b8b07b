"""Small-message all-reduce of the tensor-parallel group over peer-mapped device buffers (the library's pq_ar_*
entry points): one launch and no ring for the [tokens, hidden] partial sums of a row-parallel layer at decode
batch sizes.  Host-side counterpart of the reference's CustomAllreduce
(vllm/distributed/device_communicators/custom_all_reduce.py: IPC handle exchange over the CPU group, should_custom_ar
by size and dtype, fall back to the library collective otherwise).
"""
import ctypes
from typing import Optional

import torch
import torch.distributed as dist

from .. import _lib

_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}


def _all_agree(group, ok: bool) -> bool:
    """True when every rank of the group passed True (an object collective: works on gloo and nccl groups alike)."""
    flags = [None] * dist.get_world_size(group)
    dist.all_gather_object(flags, bool(ok), group=group)
    return all(flags)


def peers_reachable(group=None) -> bool:
    """Collective.  Whether the ranks of `group` can use each other's device memory: all on one host, and every pair
    of distinct devices with peer access (the reference's CustomAllreduce checks the same before it falls back to
    the library collective, vllm/distributed/device_communicators/custom_all_reduce.py:80-130).  Ranks that share
    one device (tests) reach each other trivially."""
    import socket
    group = group if group is not None else dist.group.WORLD
    mine = (socket.gethostname(), torch.cuda.current_device() if torch.cuda.is_available() else -1)
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, mine, group=group)
    ok = all(h == mine[0] and d >= 0 for h, d in everyone)
    if ok:
        for _, d in everyone:
            if d != mine[1]:
                try:
                    ok = ok and bool(torch.cuda.can_device_access_peer(mine[1], d))
                except Exception:
                    ok = False
    return _all_agree(group, ok)


class XgmiAllReduce:
    """Created collectively by every rank of `group` (2 to 8 ranks of one node), each with its device current.
    Every rank learns whether every other rank got its buffers mapped: on failure anywhere, all of them raise
    RuntimeError together (nobody is left waiting in a barrier).

    A wait for a peer that gives up (30 s by default) fills the call's outputs with NaN and sets a status word:
    check() -- after a stream synchronise, never under graph capture -- raises then, and so does close().  After that
    the communicator must not be used any more."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, max_bytes: Optional[int] = None):
        L = _lib.load()
        self.L = L
        self.group = group if group is not None else dist.group.WORLD
        self.rank, self.world = dist.get_rank(self.group), dist.get_world_size(self.group)
        if not 2 <= self.world <= 8:
            raise ValueError(f"XgmiAllReduce: 2 to 8 ranks, got {self.world}")
        cap = int(L.pq_ar_max_message_bytes())
        self.max_bytes = cap if max_bytes is None else min(int(max_bytes), cap)
        self.comm, self.peers, self.own = None, [], None
        own = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        err = None
        if L.pq_ar_alloc(ctypes.byref(own), handle):
            err = L.pq_ar_last_error().decode()
        else:
            self.own = own
        handles = [None] * self.world
        dist.all_gather_object(handles, None if err else handle.raw, group=self.group)
        if err is None and any(h is None for h in handles):
            err = "another rank could not allocate its buffer"
        ptrs = (ctypes.c_void_p * self.world)()
        if err is None:
            for r in range(self.world):
                if r == self.rank:
                    ptrs[r] = own.value
                    continue
                p = ctypes.c_void_p()
                if L.pq_ar_open(handles[r], ctypes.byref(p)):
                    err = L.pq_ar_last_error().decode()
                    break
                self.peers.append(p)
                ptrs[r] = p.value
        if err is None:
            comm = ctypes.c_void_p()
            if L.pq_ar_create(self.rank, self.world, ptrs, ctypes.byref(comm)):
                err = L.pq_ar_last_error().decode()
            else:
                self.comm = comm
        # every rank has mapped every buffer before the first flag is written -- or none goes on
        if not _all_agree(self.group, err is None):
            self._release()
            raise RuntimeError(f"XgmiAllReduce: {err or 'another rank failed to map the peer buffers'}")

    def should_use(self, t: torch.Tensor) -> bool:
        """Same decision on every rank: it depends on shape and dtype only."""
        n = t.numel() * t.element_size()
        return (self.comm is not None and t.is_cuda and t.dtype in _DT and t.is_contiguous() and 0 < n <= self.max_bytes
                and n % 16 == 0 and t.data_ptr() % 16 == 0)

    def all_reduce(self, t: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """SUM over the group; `out` defaults to a new tensor, `out is t` reduces in place."""
        if out is None:
            out = torch.empty_like(t)
        stream = torch.cuda.current_stream(t.device).cuda_stream
        with torch.cuda.device(t.device):
            rc = self.L.pq_ar_all_reduce(self.comm, t.data_ptr(), out.data_ptr(), t.numel(), _DT[t.dtype], stream)
        if rc:
            raise RuntimeError(self.L.pq_ar_last_error().decode())
        return out

    def add_norm_supported(self, y: torch.Tensor, residual: torch.Tensor) -> bool:
        """Whether all_reduce_add_norm takes this [tokens, hidden] pair (same decision on every rank)."""
        if self.comm is None or y.dim() < 1 or y.dtype not in (torch.float16, torch.bfloat16):
            return False
        H = y.shape[-1]
        T = y.numel() // H if H else 0
        return (y.is_cuda and y.is_contiguous() and residual.is_contiguous() and residual.shape == y.shape
                and residual.dtype == y.dtype and y.data_ptr() % 16 == 0 and residual.data_ptr() % 16 == 0
                and bool(self.L.pq_ar_add_norm_supported(T, H, _DT[y.dtype])))

    def all_reduce_add_norm(self, y: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor,
                            eps: float) -> torch.Tensor:
        """SUM of `y` over the group, + `residual` (replaced by the sum, in place), RMSNorm with `norm_weight`:
        returns the normalised tensor.  One launch; the values of all_reduce followed by fused_add_rms_norm."""
        H = y.shape[-1]
        xw = torch.empty_like(y)
        stream = torch.cuda.current_stream(y.device).cuda_stream
        with torch.cuda.device(y.device):
            rc = self.L.pq_ar_all_reduce_add_norm(self.comm, y.data_ptr(), y.numel() // H, H, _DT[y.dtype],
                                                  residual.data_ptr(), norm_weight.data_ptr(), float(eps),
                                                  xw.data_ptr(), stream)
        if rc:
            raise RuntimeError(self.L.pq_ar_last_error().decode())
        return xw

    def set_timeout_ms(self, ms: int) -> None:
        if self.L.pq_ar_set_timeout_ms(self.comm, int(ms)):
            raise RuntimeError(self.L.pq_ar_last_error().decode())

    def check(self) -> None:
        """After a synchronise: raises when a wait for a peer timed out in an earlier call (whose outputs are NaN)."""
        if self.L.pq_ar_status(self.comm):
            raise RuntimeError(self.L.pq_ar_last_error().decode())

    def _release(self) -> None:
        if self.comm is not None:
            self.L.pq_ar_destroy(self.comm)
        for p in self.peers:
            self.L.pq_ar_close(p)
        if self.own is not None:
            self.L.pq_ar_free(self.own)
        self.comm, self.peers, self.own = None, [], None

    def close(self) -> None:
        """Collective.  Raises (after releasing everything) when a wait timed out in some earlier call."""
        if self.comm is None:
            return
        torch.cuda.synchronize()
        failed = bool(self.L.pq_ar_status(self.comm))
        text = self.L.pq_ar_last_error().decode() if failed else ""
        dist.barrier(group=self.group)      # nobody still reads a buffer that is about to go
        self._release()
        if failed:
            raise RuntimeError(text)
