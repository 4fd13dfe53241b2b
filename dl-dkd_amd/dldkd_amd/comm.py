"""The communicator under the two exchanges of the one-process-per-GPU path (dist.py): all-gather of score blocks
(method/eval.py:188-212 with the gallery cut by video) and all-reduce of the flat gradient buffer (method/train.py:147-151).

GPU: RcclComm - RCCL driven directly through the C ABI (include/dldkd_hip.h, dldkd_comm_*).  Every collective is ONE enqueue on
the caller's current stream; there is no process-group object, no watchdog thread and no completion polling, so collectives
sit between hipGraph replays (and captures) of the same process like any other launch.  The rendezvous id travels over the TCP
store of torch.distributed's env:// rendezvous (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE, as torch.distributed.run sets
them); nothing else of torch.distributed is used on the GPU.
CPU tests: TorchGroupComm - the same interface over a torch.distributed group (gloo, world size 2).

`current()` is what the rest of the package asks for: the installed communicator, else the default torch.distributed group when
one is initialised (the CPU tests), else None (one process)."""
import ctypes
import os

import torch

_OPS = {"sum": 0, "max": 1, "min": 2}
_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3, torch.uint8: 4}


class _Done:
    """Handle of a collective that is already ordered on a stream (RCCL) or already complete."""

    def wait(self):
        return True


class Comm:
    rank = 0
    world = 1

    def all_reduce(self, t, op="sum", async_op=False):
        raise NotImplementedError

    def all_gather_into(self, out, inp, async_op=False):
        raise NotImplementedError

    def broadcast(self, t, src=0):
        raise NotImplementedError

    def barrier(self):
        raise NotImplementedError

    def max_over_ranks(self, value, device):
        """A host float's maximum over the ranks (bench.py's step time)."""
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        self.all_reduce(t, "max")
        return float(t.item())


class TorchGroupComm(Comm):
    """A torch.distributed group (gloo on CPU tensors in the tests) behind the same interface."""

    def __init__(self, group=None):
        import torch.distributed as tdist
        self.tdist, self.group = tdist, group
        self.rank, self.world = tdist.get_rank(group), tdist.get_world_size(group)

    def all_reduce(self, t, op="sum", async_op=False):
        ops = {"sum": self.tdist.ReduceOp.SUM, "max": self.tdist.ReduceOp.MAX, "min": self.tdist.ReduceOp.MIN}
        w = self.tdist.all_reduce(t, op=ops[op], group=self.group, async_op=async_op)
        return w if async_op else _Done()

    def all_gather_into(self, out, inp, async_op=False):
        w = self.tdist.all_gather_into_tensor(out, inp, group=self.group, async_op=async_op)
        return w if async_op else _Done()

    def broadcast(self, t, src=0):
        self.tdist.broadcast(t, src=src, group=self.group)

    def barrier(self):
        self.tdist.barrier(group=self.group)


class RcclComm(Comm):
    """One RCCL communicator of `world` ranks on `device`, created from a 128-byte id that rank 0 drew."""

    def __init__(self, world, rank, unique_id, device):
        from . import native
        self.native, self.lib = native, native.lib()
        self.device = torch.device(device)
        self.rank, self.world = int(rank), int(world)
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            native.check(self.lib.dldkd_comm_init(ctypes.byref(self._h), self.world, self.rank,
                                                  ctypes.cast(ctypes.c_char_p(unique_id), ctypes.c_void_p)), "comm_init")
        self._bar = torch.zeros(1, dtype=torch.int32, device=self.device)

    @staticmethod
    def unique_id():
        from . import native
        buf = ctypes.create_string_buffer(128)
        native.check(native.lib().dldkd_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "comm_unique_id")
        return buf.raw

    def _args(self, t):
        if not t.is_cuda or t.device != self.device:
            raise self.native.NativeError(f"RcclComm on {self.device}: tensor on {t.device}")
        if not t.is_contiguous():
            raise self.native.NativeError("RcclComm: non-contiguous tensor")
        if t.dtype not in _DTYPES:
            raise self.native.NativeError(f"RcclComm: dtype {t.dtype} not supported (fp32, fp64, int32, int64, uint8)")
        return ctypes.c_void_p(t.data_ptr()), _DTYPES[t.dtype]

    def all_reduce(self, t, op="sum", async_op=False):
        p, dt = self._args(t)
        self.native.check(self.lib.dldkd_comm_all_reduce(self._h, p, p, t.numel(), dt, _OPS[op], self.native.stream()), "comm_all_reduce")
        return _Done()

    def all_gather_into(self, out, inp, async_op=False):
        pi, dt = self._args(inp)
        po, dto = self._args(out)
        if dto != dt or out.numel() != inp.numel() * self.world:
            raise self.native.NativeError(f"all_gather_into: output {tuple(out.shape)} {out.dtype} is not world x input "
                                          f"{tuple(inp.shape)} {inp.dtype}")
        self.native.check(self.lib.dldkd_comm_all_gather(self._h, pi, po, inp.numel(), dt, self.native.stream()), "comm_all_gather")
        return _Done()

    def broadcast(self, t, src=0):
        p, dt = self._args(t)
        self.native.check(self.lib.dldkd_comm_broadcast(self._h, p, t.numel(), dt, int(src), self.native.stream()), "comm_broadcast")

    def barrier(self):
        """Every rank's current stream has reached this point: a one-word all-reduce, then the host waits for its stream."""
        self.all_reduce(self._bar, "sum")
        torch.cuda.current_stream(self.device).synchronize()
        self.check_async()

    def check_async(self):
        self.native.check(self.lib.dldkd_comm_async_error(self._h), "comm_async_error")

    def destroy(self):
        """Drain the device, then free the communicator (the ABI's destroy does not synchronise)."""
        if self._h:
            torch.cuda.synchronize(self.device)
            h, self._h = self._h, ctypes.c_void_p()
            self.native.check(self.lib.dldkd_comm_destroy(h), "comm_destroy")


_current = None


def install(c):
    """Make `c` the communicator dist.py / train.py / eval.py use (None: back to one process)."""
    global _current
    _current = c
    return c


def current():
    if _current is not None:
        return _current
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        return TorchGroupComm(None)
    return None


def info():
    """(rank, world) of the current communicator, (0, 1) without one."""
    c = current()
    return (c.rank, c.world) if c is not None else (0, 1)


def _env_store(rank, world, timeout_s=600):
    """The TCP store of the env:// rendezvous: under torch.distributed.run the agent's store, stand-alone rank 0 hosts it."""
    import datetime

    import torch.distributed as tdist
    it = tdist.rendezvous("env://", rank, world, timeout=datetime.timedelta(seconds=timeout_s))
    store, r, w = next(it)
    return store, r, w


_generation = 0


def init_rccl_from_env(device, install_default=True):
    """One communicator over all ranks of the job (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Rank 0 draws the id and
    publishes it in the store; every rank joins with it."""
    global _generation
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: the only form the pool's driver supports
    # one node (torch.distributed.run --nnodes=1, or one rank): RCCL's bootstrap sockets over loopback - the boxes have no network
    # and their hostname need not resolve; the data path is xGMI peer access either way.  An explicit setting wins.
    local = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local == world:
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    key = f"dldkd_comm_id_{_generation}"
    _generation += 1
    if world == 1:
        uid = RcclComm.unique_id()                                  # no store needed (and no port taken) for one rank
    else:
        store, rank, world = _env_store(rank, world)
        if rank == 0:
            uid = RcclComm.unique_id()
            store.set(key, uid)
        else:
            uid = bytes(store.get(key))
    c = RcclComm(world, rank, uid, device)
    return install(c) if install_default else c
