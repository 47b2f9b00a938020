"""Thin handle over the C-ABI collectives (include/lmod_hip.h, csrc/comm.hip): RCCL bound at run time inside the kernel
library, one communicator per process.  The training engine (llavamod.engine) exchanges through torch.distributed — the
same RCCL — so this module is the reference binding of the boundary for hosts that do not use torch.distributed, and what
tests/test_comm_gpu.py drives."""
import ctypes

import torch

from . import _hip

_DT = {torch.float32: 0, torch.bfloat16: 1}


def unique_id():
    """128-byte communicator id (rank 0 creates it; distribute it to the other ranks)."""
    buf = ctypes.create_string_buffer(128)
    rc = _hip.load().lmod_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError(f"lmod_comm_unique_id failed: {_hip._ERR.get(rc, rc)}")
    return buf.raw


class NativeComm:
    def __init__(self, uid: bytes, rank: int, world: int):
        self.rank, self.world = rank, world
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(uid, 128)
        rc = _hip.load().lmod_comm_init(ctypes.cast(ctypes.byref(h), ctypes.c_void_p), ctypes.cast(buf, ctypes.c_void_p), rank, world)
        if rc != 0:
            raise RuntimeError(f"lmod_comm_init failed: {_hip._ERR.get(rc, rc)}")
        self._h = h

    def close(self):
        if self._h is not None:
            _hip.load().lmod_comm_destroy(self._h)
            self._h = None

    def allreduce_(self, buf):
        _hip.call("lmod_allreduce_grads", self._h, _hip.ptr(buf), buf.numel(), _DT[buf.dtype])
        return buf

    def reduce_scatter_(self, span):
        """span [world * n] -> this rank's chunk of the sum, in place; returns the chunk view."""
        n = span.numel() // self.world
        _hip.call("lmod_reduce_scatter_grads", self._h, _hip.ptr(span), n, _DT[span.dtype])
        return span[self.rank * n:(self.rank + 1) * n]

    def allgather_(self, span):
        n = span.numel() // self.world
        _hip.call("lmod_allgather_params", self._h, _hip.ptr(span), n, _DT[span.dtype])
        return span

    def moe_all_to_all(self, send, send_rows, recv_rows):
        """send [sum(send_rows), H] bf16 packed live rows -> recv [sum(recv_rows), H]."""
        H = send.shape[1]
        recv = torch.empty((int(sum(recv_rows)), H), device=send.device, dtype=torch.bfloat16)
        A = ctypes.c_longlong * self.world
        sr, rr = A(*[int(x) for x in send_rows]), A(*[int(x) for x in recv_rows])
        _hip.call("lmod_moe_all_to_all", self._h, _hip.ptr(send), _hip.ptr(recv), ctypes.cast(sr, ctypes.c_void_p),
                  ctypes.cast(rr, ctypes.c_void_p), H)
        return recv


# ONE communicator per process over the world: the engine's gradient exchange (side stream) and the expert-parallel
# all-to-all (compute stream) share it — two RCCL communicators with collectives in flight on different streams is the classic
# deadlock hazard, one communicator serialises its operations in issue order, which every rank follows identically.
_SHARED = None


def shared_world_comm():
    """The process's C-ABI communicator over torch.distributed's world, created on first use (COLLECTIVE: rank 0 draws the id, everybody
    receives it over the existing process group — call it from a point every rank reaches together, e.g. DataParallel.attach)."""
    global _SHARED
    if _SHARED is None:
        import atexit
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = [unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        _SHARED = NativeComm(uid[0], rank, world)
        atexit.register(close_shared)
    return _SHARED


def close_shared():
    """Destroy the shared communicator (teardown; idempotent)."""
    global _SHARED
    if _SHARED is not None:
        try:
            _SHARED.close()
        finally:
            _SHARED = None
