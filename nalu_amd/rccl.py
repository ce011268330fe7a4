"""Loader of libnalu_gpu_rccl.so (include/nalu_gpu_rccl.h): the comm callbacks of the library
natively on RCCL, so that no Python runs per Krylov iteration.  torch.distributed is only the
bootstrap here (it carries the 128-byte unique id from rank 0 to the others, MPI_Bcast's role);
`nalu_amd.comm.TorchComm` stays for the gloo tests (several ranks on one GPU)."""
import ctypes as C
import os

from . import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
RCCL_LIB_PATH = os.environ.get("NALU_GPU_RCCL_LIB", os.path.join(_HERE, "libnalu_gpu_rccl.so"))
ID_BYTES = 128
SYMBOLS = ["nalu_gpu_rccl_get_unique_id", "nalu_gpu_rccl_world_create", "nalu_gpu_rccl_world_destroy",
           "nalu_gpu_rccl_attach", "nalu_gpu_rccl_attach_flags", "nalu_gpu_rccl_detach", "nalu_gpu_rccl_last_error"]
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    L.load()  # libnalu_gpu.so (and torch's HIP / RCCL runtime) first
    if not os.path.exists(RCCL_LIB_PATH):
        raise L.NaluGpuError(f"{RCCL_LIB_PATH} is missing: build it with `make -C nalu_amd/csrc`")
    lib = C.CDLL(RCCL_LIB_PATH)
    vp, ci = C.c_void_p, C.c_int
    lib.nalu_gpu_rccl_last_error.restype = C.c_char_p
    for name, args in {"nalu_gpu_rccl_get_unique_id": [vp],
                       "nalu_gpu_rccl_world_create": [vp, ci, ci, ci, C.POINTER(vp)],
                       "nalu_gpu_rccl_world_destroy": [vp],
                       "nalu_gpu_rccl_attach": [vp, vp, C.POINTER(vp)],
                       "nalu_gpu_rccl_attach_flags": [vp, vp, ci, C.POINTER(vp)],
                       "nalu_gpu_rccl_detach": [vp]}.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = ci
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise L.NaluGpuError(f"nalu_gpu_rccl error {rc}: {load().nalu_gpu_rccl_last_error().decode()}")


class RcclWorld:
    """One RCCL communicator per process (one process per GPU)."""

    def __init__(self, rank, n_ranks, device_index, broadcast_id=None):
        """broadcast_id(bytes_or_None) -> bytes: hands rank 0's id to every rank; defaults to
        torch.distributed.broadcast_object_list on the default process group."""
        lib = load()
        buf = C.create_string_buffer(ID_BYTES)
        if rank == 0:
            _check(lib.nalu_gpu_rccl_get_unique_id(buf))
        if n_ranks > 1:
            if broadcast_id is None:
                import torch.distributed as dist

                def broadcast_id(b):
                    box = [b]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            got = broadcast_id(buf.raw if rank == 0 else None)
            buf = C.create_string_buffer(got, ID_BYTES)
        self.w = C.c_void_p()
        _check(lib.nalu_gpu_rccl_world_create(buf, rank, n_ranks, device_index, C.byref(self.w)))
        self.rank, self.n_ranks = rank, n_ranks
        self._comms = []

    def attach(self, linsys, allow_self_neighbour=False):
        """Install the RCCL callbacks on a finalized GpuLinearSystem whose halo plan is set.
        allow_self_neighbour: a halo neighbour may be this rank (a mesh periodic across its own boundary;
        the loop-back test of the one-GPU box)."""
        c = C.c_void_p()
        _check(load().nalu_gpu_rccl_attach_flags(self.w, linsys.h, 1 if allow_self_neighbour else 0, C.byref(c)))
        self._comms.append(c)
        return c

    def detach(self, c):
        if c in self._comms:
            self._comms.remove(c)
            load().nalu_gpu_rccl_detach(c)

    def close(self):
        for c in list(self._comms):
            self.detach(c)
        if self.w:
            load().nalu_gpu_rccl_world_destroy(self.w)
            self.w = C.c_void_p()
