"""RCCL communicator behind the C-ABI (vitk_comm_* in include/vitk.h).

The unique id travels from rank 0 to the other ranks through ``torch.distributed`` (any initialised
backend); after that the gradient exchange itself runs through ``vitk_allreduce`` / ``vitk_broadcast``
on the stream the caller names.  `GradBucketer(transport="rccl")` uses it; the default transport
stays ``torch.distributed``'s own RCCL binding.
"""
import ctypes

import torch
import torch.distributed as dist

from . import check, dtype_code, lib

UNIQUE_ID_BYTES = 128


class Communicator:
    def __init__(self, rank=None, world=None, device=None):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        if self.rank == 0:
            check(lib.vitk_comm_unique_id(buf, UNIQUE_ID_BYTES), "vitk_comm_unique_id")
        box = [bytes(buf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        self._id = ctypes.create_string_buffer(box[0], UNIQUE_ID_BYTES)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.vitk_comm_init(self.rank, self.world, self._id, ctypes.byref(handle)), "vitk_comm_init")
        self._comm = handle

    def _stream(self, stream):
        return (stream or torch.cuda.current_stream(self.device)).cuda_stream

    def all_reduce(self, t, average=True, stream=None):
        """In place over a contiguous CUDA tensor, enqueued on `stream` (default: current)."""
        assert t.is_cuda and t.is_contiguous()
        check(lib.vitk_allreduce(self._comm, t.data_ptr(), t.numel(), dtype_code(t.dtype), int(average),
                                 self._stream(stream)), "vitk_allreduce")
        return t

    def broadcast(self, t, root=0, stream=None):
        assert t.is_cuda and t.is_contiguous()
        check(lib.vitk_broadcast(self._comm, t.data_ptr(), t.numel(), dtype_code(t.dtype), root,
                                 self._stream(stream)), "vitk_broadcast")
        return t

    def close(self):
        if self._comm:
            check(lib.vitk_comm_destroy(self._comm), "vitk_comm_destroy")
            self._comm = None
