"""Single-process collective over the GPUs of one node: ctypes face of rtx_comm_init_all / rtx_allgather /
rtx_comm_destroy (include/radtxfr_hip.h). One host process, several devices, no launcher -- the way the reference's own
scripts are run (Generate_LWIR_TUD.py is a plain program). RCCL (ncclCommInitAll + grouped ncclAllGather over xGMI) when
the devices are distinct and librccl loads, else every device pulls its peers' blocks with hipMemcpyPeerAsync."""
import ctypes as C

import torch

from . import _lib


class LocalComm:
    def __init__(self, devices, backend=-1):
        """devices: GPU indices, one per rank (an index may repeat with the peer-copy backend: ranks sharing a device).
        backend -1 auto, 0 peer copies, 1 RCCL."""
        self.devices = [int(d) for d in devices]
        self._lib = _lib.load()
        self._h = C.c_void_p(0)
        arr = (C.c_int * len(self.devices))(*self.devices)
        _lib.check(self._lib.rtx_comm_init_all(len(self.devices), arr, int(backend), C.byref(self._h)))

    @property
    def backend(self):
        return "rccl" if self._lib.rtx_comm_backend(self._h) == 1 else "peer"

    def all_gather(self, send, recv, streams=None):
        """send[i]: float32 contiguous tensor of `count` elements on devices[i]; recv[i]: float32 tensor of
        len(devices) * count elements on devices[i]. streams[i]: rank i's torch stream (default: its device's current
        stream). Asynchronous: the gathered blocks are complete in the order of each rank's stream."""
        n = len(self.devices)
        assert len(send) == len(recv) == n
        count = send[0].numel()
        for i in range(n):
            assert send[i].dtype == recv[i].dtype == torch.float32 and send[i].is_contiguous() and recv[i].is_contiguous()
            assert send[i].numel() == count and recv[i].numel() == n * count
            assert send[i].device.index == self.devices[i] == recv[i].device.index
        if streams is None:
            streams = [torch.cuda.current_stream(d) for d in self.devices]
        vp = C.c_void_p * n
        _lib.check(self._lib.rtx_allgather(self._h, vp(*[t.data_ptr() for t in send]), vp(*[t.data_ptr() for t in recv]), count,
                                           vp(*[s.cuda_stream for s in streams])))

    def close(self):
        if self._h:
            self._lib.rtx_comm_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
