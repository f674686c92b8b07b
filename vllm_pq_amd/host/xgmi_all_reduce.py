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


class XgmiAllReduce:
    """Created collectively by every rank of `group` (2 to 8 ranks of one node), each with its device current."""

    def __init__(self, group: Optional["dist.ProcessGroup"] = None, max_bytes: Optional[int] = None):
        L = _lib.load()
        self.L = L
        self.group = group if group is not None else dist.group.WORLD
        self.rank, self.world = dist.get_rank(self.group), dist.get_world_size(self.group)
        if not 2 <= self.world <= 8:
            raise ValueError(f"XgmiAllReduce: 2 to 8 ranks, got {self.world}")
        cap = int(L.pq_ar_max_message_bytes())
        self.max_bytes = cap if max_bytes is None else min(int(max_bytes), cap)
        own = ctypes.c_void_p()
        handle = ctypes.create_string_buffer(64)
        if L.pq_ar_alloc(ctypes.byref(own), handle):
            raise RuntimeError(L.pq_ar_last_error().decode())
        self.own = own
        handles = [None] * self.world
        dist.all_gather_object(handles, handle.raw, group=self.group)
        self.peers = []
        ptrs = (ctypes.c_void_p * self.world)()
        for r in range(self.world):
            if r == self.rank:
                ptrs[r] = own.value
            else:
                p = ctypes.c_void_p()
                if L.pq_ar_open(handles[r], ctypes.byref(p)):
                    raise RuntimeError(L.pq_ar_last_error().decode())
                self.peers.append(p)
                ptrs[r] = p.value
        comm = ctypes.c_void_p()
        if L.pq_ar_create(self.rank, self.world, ptrs, ctypes.byref(comm)):
            raise RuntimeError(L.pq_ar_last_error().decode())
        self.comm = comm
        dist.barrier(group=self.group)      # every rank has mapped every buffer before the first flag is written

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

    def check(self) -> None:
        """After a synchronise: raises when a wait for a peer timed out in an earlier call."""
        if self.L.pq_ar_status(self.comm):
            raise RuntimeError(self.L.pq_ar_last_error().decode())

    def close(self) -> None:
        if self.comm is None:
            return
        torch.cuda.synchronize()
        dist.barrier(group=self.group)      # nobody still reads a buffer that is about to go
        self.L.pq_ar_destroy(self.comm)
        for p in self.peers:
            self.L.pq_ar_close(p)
        self.L.pq_ar_free(self.own)
        self.comm, self.peers = None, []
