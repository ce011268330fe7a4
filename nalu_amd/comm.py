"""Multi-GPU plumbing: the three callbacks of nalu_gpu_comm implemented with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node; "gloo" in tests).

The library packs boundary values into its send buffer and reduces dot products into its
reduce buffer on the device; this module only moves those bytes between ranks:
  * halo exchange  = the Tpetra Import inside CrsMatrix::apply (src/LinearSolver.C:164,224)
                     and stk::mesh::copy_owned_to_shared (src/LinearSystem.C:115)
  * all-reduce     = Tpetra dot / norm2 (src/LinearSolver.C:172, src/TpetraLinearSystem.C:1756)
"""
import contextlib
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L


class _DevArray:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can wrap it."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 2, "strides": None}


def wrap_device(ptr, n, device):
    if n == 0 or not ptr:
        return torch.zeros(0, dtype=torch.float64, device=device)
    return torch.as_tensor(_DevArray(ptr, n), device=device)


def exchange_halo(send, recv, neighbors, send_offsets, recv_offsets, group=None):
    """Post the neighbour sends/receives of one halo exchange; returns the work handles.
    send[so[p]:so[p+1]] goes to neighbour p, recv[ro[p]:ro[p+1]] comes from it."""
    ops = []
    for n, p in enumerate(neighbors):
        p = int(p)
        a, b = int(send_offsets[n]), int(send_offsets[n + 1])
        if b > a:
            ops.append(dist.P2POp(dist.isend, send[a:b], p, group=group))
        a, b = int(recv_offsets[n]), int(recv_offsets[n + 1])
        if b > a:
            ops.append(dist.P2POp(dist.irecv, recv[a:b], p, group=group))
    return dist.batch_isend_irecv(ops) if ops else []


class TorchComm:
    """Owns the ctypes callback struct for one GpuLinearSystem."""

    def __init__(self, linsys, neighbors, send_offsets, recv_offsets, device, group=None):
        self.group = group
        self.device = device
        self.neighbors = [int(p) for p in neighbors]
        self.so = [int(v) for v in send_offsets]
        self.ro = [int(v) for v in recv_offsets]
        (sb, sc), (rb, rc), (red, cap) = linsys.comm_buffers()
        self.send = wrap_device(sb, sc, device)
        self.recv = wrap_device(rb, rc, device)
        self.reduce = wrap_device(red, cap, device)
        self.pending = []
        self.errors = []
        self._streams = {}
        self.linsys_lib = linsys.lib
        # gloo (tests: several ranks on one GPU) has no device send/recv: stage through the host
        self.stage = dist.get_backend(group) == "gloo" and str(device).startswith("cuda")
        self._recv_host = None
        # the P2P op list and the reduce-buffer views are built once: the callbacks run two to
        # three times per Krylov iteration and their host cost must stay below the kernels'
        self._ops = None if self.stage else self._make_ops(self.send, self.recv)
        self._views = {}
        self._ar = L.ALLREDUCE_FN(self._allreduce)
        self._xb = L.EXCHANGE_FN(self._exchange_begin)
        self._xe = L.EXCHANGE_FN(self._exchange_end)
        self._xr = L.EXCHANGE_ROWS_FN(self._exchange_rows)
        self.struct = L.Comm(None, self._ar, self._xb, self._xe, self._xr)
        linsys.set_comm(self.struct)

    def _make_ops(self, send, recv):
        ops = []
        for n, p in enumerate(self.neighbors):
            a, b = self.so[n], self.so[n + 1]
            if b > a:
                ops.append(dist.P2POp(dist.isend, send[a:b], p, group=self.group))
            a, b = self.ro[n], self.ro[n + 1]
            if b > a:
                ops.append(dist.P2POp(dist.irecv, recv[a:b], p, group=self.group))
        return ops

    def _on(self, stream):
        """Context that makes the library's stream (the callback's `stream` argument) torch's
        current stream: torch.distributed orders its collectives against the CURRENT stream, and
        the library has enqueued the pack / partial-sum kernels on ITS stream, which need not be
        the one that is current when the callback runs (GpuLinearSystem(stream=...))."""
        if not str(self.device).startswith("cuda"):
            return contextlib.nullcontext()
        key = int(stream or 0)
        ts = self._streams.get(key)
        if ts is None:
            ts = (torch.cuda.ExternalStream(key, device=self.device) if key
                  else torch.cuda.default_stream(self.device))  # hipStream_t 0 = the default stream
            self._streams[key] = ts
        return torch.cuda.stream(ts)

    def _fail(self, e):
        self.errors.append(repr(e))
        self.linsys_lib.nalu_gpu_set_callback_error(repr(e).encode()[:900])
        return 1

    # callbacks must never raise through ctypes: record and return non-zero
    def _allreduce(self, ctx, count, stream):
        try:
            with self._on(stream):
                if self.stage:
                    t = self.reduce[:count].cpu()  # on the library's stream, then host-synchronous
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                    self.reduce[:count].copy_(t)
                else:
                    v = self._views.get(count)
                    if v is None:
                        v = self._views[count] = self.reduce[:count]
                    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception as e:  # pragma: no cover
            return self._fail(e)

    def _exchange_begin(self, ctx, stream):
        try:
            with self._on(stream):
                if self.stage:
                    send = self.send.cpu()
                    recv = self._recv_host = torch.empty(self.recv.shape, dtype=torch.float64)
                    self.pending = exchange_halo(send, recv, self.neighbors, self.so, self.ro, self.group)
                else:
                    self.pending = dist.batch_isend_irecv(self._ops) if self._ops else []
            return 0
        except Exception as e:  # pragma: no cover
            return self._fail(e)

    def _exchange_end(self, ctx, stream):
        try:
            with self._on(stream):
                for w in self.pending:
                    w.wait()  # nccl: makes the current (= the library's) stream wait for the data
                self.pending = []
                if self.stage:
                    self.recv.copy_(self._recv_host)
            return 0
        except Exception as e:  # pragma: no cover
            return self._fail(e)

    def _exchange_rows(self, ctx, send_ptr, send_off, recv_ptr, recv_off, stream):
        """loadComplete's row export (export mode): caller-given device buffers, offsets per neighbour"""
        try:
            with self._on(stream):
                nn = len(self.neighbors)
                so = [int(send_off[p]) for p in range(nn + 1)]
                ro = [int(recv_off[p]) for p in range(nn + 1)]
                send = wrap_device(send_ptr, so[-1], self.device)
                recv = wrap_device(recv_ptr, ro[-1], self.device)
                if self.stage:
                    hs, hr = send.cpu(), torch.empty(ro[-1], dtype=torch.float64)
                    for w in exchange_halo(hs, hr, self.neighbors, so, ro, self.group):
                        w.wait()
                    if ro[-1]:
                        recv.copy_(hr)
                else:
                    for w in exchange_halo(send, recv, self.neighbors, so, ro, self.group):
                        w.wait()
            return 0
        except Exception as e:  # pragma: no cover
            return self._fail(e)
