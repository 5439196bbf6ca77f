"""A raw RCCL communicator for callers that drive the exchange step through the C ABI (``btsbot_allreduce_grads``)
instead of ``torch.distributed`` -- what a C / C++ host binding only ``libbtsbot_hip.so`` does with its own
``ncclCommInitRank``.  ``Trainer(..., rccl_comm=RcclComm(...))`` then needs no process group for the gradients.

The library resolves RCCL by soname at its first collective; this module loads the same way (``librccl.so.1``: the copy
PyTorch has already mapped wins), so the communicator and the ``ncclAllReduce`` that uses it come from one library.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * 128)]


def _load():
    last = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL)
        except OSError as e:        # noqa: PERF203
            last = e
    raise OSError(f"btsbot_amd.rccl: cannot load RCCL: {last}")


class RcclComm:
    """``RcclComm(rank, world, unique_id)``: rank 0 calls ``RcclComm.unique_id()`` and ships the 128 bytes to the
    other ranks (any channel); every rank then constructs its communicator with the same id, with its GPU current."""

    def __init__(self, rank: int, world: int, unique_id: bytes):
        self._lib = _load()
        uid = _UniqueId()
        C.memmove(C.byref(uid), unique_id, 128)
        self._comm = C.c_void_p()
        self._lib.ncclCommInitRank.restype = C.c_int
        self._lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        rc = self._lib.ncclCommInitRank(C.byref(self._comm), world, uid, rank)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank returned {rc}")
        self.rank, self.world = rank, world

    @staticmethod
    def unique_id() -> bytes:
        lib = _load()
        uid = _UniqueId()
        lib.ncclGetUniqueId.restype = C.c_int
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        rc = lib.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId returned {rc}")
        return bytes(uid.internal)

    @property
    def ptr(self) -> Optional[int]:
        return self._comm.value

    def destroy(self):
        if self._comm:
            self._lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self._lib.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()
