"""The RCCL communicator of libpre3 (include/pre3.h, "the RCCL communicator"): with one attached, a sharded RANSAC round
(EkfFilter.ransac_sharded_stream) and a sharded match (MatchShard.match) run scoring / collective / selection back to back on the
library's own stream -- no host synchronisation either side of the collective, one wait per round.

One process per GPU.  The 128-byte id made on rank 0 reaches the other ranks through whatever the host program has; here that is
torch.distributed (any backend: the id is bytes).  `3pre_amd.dist` keeps the torch.distributed form of the same two stages (gloo in the
CPU tests, and the fallback when RCCL cannot be bound)."""
import ctypes as C

import numpy as np

from ._lib import lib, check

ID_BYTES = 128


def unique_id():
    buf = (C.c_ubyte * ID_BYTES)()
    check(lib.pre3_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """pre3_comm: create() is collective over `world` ranks (world = 1: on its own)."""

    def __init__(self, device, id_bytes, rank, world):
        if len(id_bytes) != ID_BYTES:
            raise ValueError("a communicator id is %d bytes" % ID_BYTES)
        self._h = C.c_void_p()
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        check(lib.pre3_comm_create(C.byref(self._h), int(device), C.c_char_p(id_bytes), int(rank), int(world)))

    @classmethod
    def from_torch_distributed(cls, device):
        """the id travels as a broadcast over the process group that torch.distributed.run set up (rank 0 makes it); without a process group:
        a communicator of one rank"""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls(device, unique_id(), 0, 1)
        rank, world = dist.get_rank(), dist.get_world_size()
        on_gpu = dist.get_backend() == "nccl"
        t = torch.zeros(ID_BYTES, dtype=torch.uint8, device=torch.device("cuda", int(device)) if on_gpu else "cpu")
        if rank == 0:
            t.copy_(torch.from_numpy(np.frombuffer(unique_id(), np.uint8).copy()))
        dist.broadcast(t, src=0)
        return cls(device, bytes(t.cpu().numpy().tobytes()), rank, world)

    def info(self):
        r, w, v = C.c_int(0), C.c_int(0), C.c_int(0)
        path = C.create_string_buffer(512)
        check(lib.pre3_comm_info(self._h, C.byref(r), C.byref(w), C.byref(v), path, 512))
        return dict(rank=r.value, world=w.value, rccl_version=v.value, library=path.value.decode(errors="replace"))

    def set_timeout(self, milliseconds):
        """deadline of the host waits behind this communicator's collectives (default 10 s): on expiry the communicator is aborted and the call
        returns PRE3_E_COMM"""
        check(lib.pre3_comm_set_timeout(self._h, int(milliseconds)))

    def close(self):
        if getattr(self, "_h", None) and lib is not None:          # (lib is None while the interpreter shuts down)
            lib.pre3_comm_destroy(self._h)
            self._h = None

    __del__ = close
