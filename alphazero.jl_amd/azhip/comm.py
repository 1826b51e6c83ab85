"""The multi-GPU exchange of the path (az_comm_*, csrc/comm.hip): RCCL over xGMI behind the C ABI.

simulate_distributed (src/simulations.jl:252-290) on one process per GPU: every rank simulates its shard of the games
(global game ids, no communication), then `Comm.gather_push` all-gathers the device-resident records of all ranks and
pushes every game into the rank's device replay memory; `Comm.broadcast_params` ships a new network.  No torch in the
data path: the 128-byte unique id is the only thing that travels by the host's own means (here: any callable that
broadcasts bytes from rank 0, e.g. torch.distributed's object broadcast, MPI, a shared file)."""
import ctypes as C

from . import _lib as L


def unique_id():
    buf = (C.c_uint8 * L.COMM_ID_BYTES)()
    L.check(L.lib().az_comm_unique_id(buf))
    return bytes(buf)


def version():
    """(ncclGetVersion code, path of the collective library az_comm_* bound)"""
    v = C.c_int32()
    buf = C.create_string_buffer(512)
    L.check(L.lib().az_comm_version(C.byref(v), buf, 512))
    return int(v.value), buf.value.decode("utf-8", "replace")


class Comm:
    def __init__(self, device, rank, world, uid):
        assert len(uid) == L.COMM_ID_BYTES
        h = C.c_void_p()
        buf = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(uid)
        L.check(L.lib().az_comm_init(device, rank, world, buf, C.byref(h)))
        self._h, self.rank, self.world, self.device = h, rank, world, device

    def close(self):
        if getattr(self, "_h", None):
            L.lib().az_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def gather_push(self, engine, memory, gamma=1.0):
        """collective: all ranks' phase records -> (optionally) this rank's MemoryBuffer; returns GatherStats"""
        st = L.GatherStats()
        L.check(L.lib().az_comm_gather_push(self._h, engine._h, memory._h if memory is not None else None, float(gamma), C.byref(st)))
        return st

    def broadcast_params(self, engine, root=0):
        L.check(L.lib().az_comm_broadcast_params(self._h, engine._h, root))


def torch_broadcast_id(rank):
    """unique id from rank 0 to every rank of the default torch.distributed group (any backend)"""
    import torch
    import torch.distributed as dist
    box = [unique_id() if rank == 0 else None]
    if dist.get_backend() == "nccl":
        dist.broadcast_object_list(box, src=0, device=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.broadcast_object_list(box, src=0)
    return box[0]
