"""RCCL through the C ABI of libtimer1_hip.so (include/timer1_hip.h: tr1_rccl_*), for hosts that do not go through torch.distributed.

`time-r1_amd/dist.py` (the trainer's data-parallel path) uses torch.distributed's "nccl" backend, which is the same librccl; this module is the
thin binding of SURVEY 8b's `rccl_{init, allreduce, reduce_scatter, allgather}` row: one communicator per process (one process per GPU), every
collective a SUM, asynchronous on the caller's HIP stream.  The 128-byte unique id is created by rank 0 and distributed by the launcher (a file,
an environment variable, a TCP store) - `RcclComm.from_env()` reads it from TR1_RCCL_ID_FILE."""
import ctypes
import os
import time

import torch

from . import hip

_DT = {torch.bfloat16: 0, torch.float32: 1, torch.int32: 2}


def version():
    return int(hip.lib().raw("tr1_rccl_version")())


def unique_id():
    buf = ctypes.create_string_buffer(128)
    hip.lib().call("tr1_rccl_unique_id", ctypes.cast(buf, ctypes.c_void_p))
    return buf.raw


class RcclComm:
    def __init__(self, uid: bytes, world: int, rank: int, device=None):
        assert len(uid) == 128 and 0 <= rank < world
        if device is not None:
            torch.cuda.set_device(device)
        self.world, self.rank = world, rank
        slot = ctypes.c_void_p()
        ub = ctypes.create_string_buffer(uid, 128)
        hip.lib().call("tr1_rccl_init", ctypes.cast(ub, ctypes.c_void_p), world, rank, ctypes.cast(ctypes.pointer(slot), ctypes.c_void_p))
        self.comm = slot.value

    @classmethod
    def from_env(cls, device=None, timeout_s=120.0):
        """RANK / WORLD_SIZE as torchrun sets them; rank 0 writes the id to TR1_RCCL_ID_FILE, the others wait for it."""
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        path = os.environ.get("TR1_RCCL_ID_FILE", "/tmp/tr1_rccl_id.%s" % os.environ.get("MASTER_PORT", "0"))
        if rank == 0:
            uid = unique_id()
            with open(path + ".tmp", "wb") as f:
                f.write(uid)
            os.replace(path + ".tmp", path)
        else:
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > timeout_s:
                    raise TimeoutError("no RCCL unique id at %s after %.0f s" % (path, timeout_s))
                time.sleep(0.05)
            uid = open(path, "rb").read()
        return cls(uid, world, rank, device)

    def _s(self):
        return torch.cuda.current_stream().cuda_stream

    def all_reduce_(self, t):
        assert t.is_contiguous() and t.is_cuda
        hip.lib().call("tr1_rccl_allreduce", self.comm, t.data_ptr(), t.data_ptr(), t.numel(), _DT[t.dtype], self._s())
        return t

    def reduce_scatter(self, out, inp):
        assert out.is_contiguous() and inp.is_contiguous() and inp.numel() == out.numel() * self.world and out.dtype == inp.dtype
        hip.lib().call("tr1_rccl_reduce_scatter", self.comm, inp.data_ptr(), out.data_ptr(), out.numel(), _DT[out.dtype], self._s())
        return out

    def all_gather(self, out, inp):
        assert out.is_contiguous() and inp.is_contiguous() and out.numel() == inp.numel() * self.world and out.dtype == inp.dtype
        hip.lib().call("tr1_rccl_allgather", self.comm, inp.data_ptr(), out.data_ptr(), inp.numel(), _DT[inp.dtype], self._s())
        return out

    def close(self):
        if self.comm:
            hip.lib().call("tr1_rccl_destroy", self.comm)
            self.comm = None
