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
_PROC_START = time.time()
_STALE_S = 300.0      # ranks of one job start within minutes of each other: an id file older than that belongs to an earlier (crashed) job


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
        # The id travels through a 0600 file in a 0700 directory owned by this user (not a predictable world-writable path), tagged with the job's
        # nonce (torchrun's run id + master port); rank 0 removes what an earlier job left before it writes, and removes its own file after the
        # communicator is up.  Readers refuse symlinks, other users' files, another job's tag and files older than their own start.
        path = os.environ.get("TR1_RCCL_ID_FILE")
        if path is None:
            d = os.path.join(os.environ.get("XDG_RUNTIME_DIR") or "/tmp", "tr1_rccl.%d" % os.getuid())
            os.makedirs(d, mode=0o700, exist_ok=True)
            st = os.lstat(d)
            if not os.path.isdir(d) or os.path.islink(d) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
                raise RuntimeError("%s must be a private directory of this user (set TR1_RCCL_ID_FILE to choose another place)" % d)
            path = os.path.join(d, "id.%s" % os.environ.get("MASTER_PORT", "0"))
        nonce = ("%s|%s|" % (os.environ.get("TORCHELASTIC_RUN_ID", ""), os.environ.get("MASTER_PORT", ""))).encode()
        t_start = _PROC_START
        if rank == 0:
            for p in (path, path + ".tmp"):
                try:
                    os.unlink(p)
                except FileNotFoundError:
                    pass
            uid = unique_id()
            fd = os.open(path + ".tmp", os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "wb") as f:
                f.write(nonce + uid)
            os.replace(path + ".tmp", path)
        else:
            t0 = time.time()
            uid = None
            while uid is None:
                try:
                    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                    with os.fdopen(fd, "rb") as f:
                        st = os.fstat(f.fileno())
                        blob = f.read()
                    if st.st_uid == os.getuid() and st.st_mtime >= t_start - _STALE_S and blob.startswith(nonce) and len(blob) > len(nonce):
                        uid = blob[len(nonce):]
                except (FileNotFoundError, OSError):
                    pass
                if uid is None:
                    if time.time() - t0 > timeout_s:
                        raise TimeoutError("no fresh RCCL unique id at %s after %.0f s" % (path, timeout_s))
                    time.sleep(0.05)
        comm = cls(uid, world, rank, device)      # ncclCommInitRank returns once every rank has joined, i.e. has read the file
        if rank == 0:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
        return comm

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
