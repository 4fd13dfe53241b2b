"""The communicator under the two exchanges of the one-process-per-GPU path (dist.py): all-gather of score blocks
(method/eval.py:188-212 with the gallery cut by video) and all-reduce of the flat gradient buffer (method/train.py:147-151).

GPU: RcclComm - RCCL driven directly through the C ABI (include/dldkd_hip.h, dldkd_comm_*).  Every collective is ONE enqueue on
the caller's current stream; there is no process-group object, no watchdog thread and no completion polling, so collectives
sit between hipGraph replays (and captures) of the same process like any other launch.  The rendezvous id travels over the TCP
store of torch.distributed's env:// rendezvous (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE, as torch.distributed.run sets
them); nothing else of torch.distributed is used on the GPU.
CPU tests: TorchGroupComm - the same interface over a torch.distributed group (gloo, world size 2).

`current()` is what the rest of the package asks for: the installed communicator, else the default torch.distributed group when
one is initialised (the CPU tests), else None (one process).

Hang detection (round 6).  There is no watchdog thread (the r04 abort WAS torch's watchdog), so a dead or
diverged peer would park this rank's stream inside a collective for ever and the next host read (`float(loss)`, `.cpu()`,
`barrier`) with it.  Every place where the host of the multi-rank path already blocks goes through `Comm.host_wait()` instead of
a bare synchronise: it records an event behind the work, polls it (`hipEventQuery`) together with the communicator's asynchronous
error state (`dldkd_comm_async_error`) IN THE CALLING THREAD, and at `deadline_s` (DLDKD_COMM_DEADLINE_S, default 300 s) aborts
the communicator (`dldkd_comm_abort`: RCCL's kernels poll the abort flag and leave the stream) and raises CommTimeout - on every
rank that is still running, so `torch.distributed.run` sees non-zero exits and takes the job down."""
import ctypes
import os
import time

import torch


class CommTimeout(RuntimeError):
    """A peer did not arrive at a collective (or the work in front of it never finished) within the deadline."""


def _default_deadline():
    try:
        return float(os.environ.get("DLDKD_COMM_DEADLINE_S", "300"))
    except ValueError:
        return 300.0

_OPS = {"sum": 0, "max": 1, "min": 2}
_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3, torch.uint8: 4}


class _Done:
    """Handle of a collective that is already ordered on a stream (RCCL) or already complete."""

    def wait(self):
        return True


class Comm:
    rank = 0
    world = 1
    deadline_s = None           # None: DLDKD_COMM_DEADLINE_S at the time of the wait (default 300 s); <= 0: wait for ever

    def _deadline(self, deadline_s=None):
        d = deadline_s if deadline_s is not None else (self.deadline_s if self.deadline_s is not None else _default_deadline())
        return float(d)

    def host_wait(self, stream=None, what="", deadline_s=None):
        """The host waits - deadline-bounded - until everything enqueued so far on `stream` (default: the current one) is done."""
        raise NotImplementedError

    def watch(self, stream=None, what=""):
        """Progress check that does not drain the pipeline (a replayed training loop that never reads the loss back): the marker
        left by the PREVIOUS call must be complete by now - else it is waited for against the deadline - and a new one is left."""
        return True

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
        self.host_wait(what="max_over_ranks")
        return float(t.item())


class _PolledWork:
    """A torch.distributed work handle whose wait() is deadline-bounded (TorchGroupComm)."""

    def __init__(self, comm, work, what):
        self.comm, self.work, self.what = comm, work, what

    def wait(self):
        self.comm._poll(self.work, self.what)
        return True


class TorchGroupComm(Comm):
    """A torch.distributed group (gloo on CPU tensors in the tests) behind the same interface.  Every collective is issued
    asynchronously and its completion polled against the deadline in the calling thread (gloo's own blocking wait has a
    30-minute default and no way to say which collective hung)."""

    def __init__(self, group=None):
        import torch.distributed as tdist
        self.tdist, self.group = tdist, group
        self.rank, self.world = tdist.get_rank(group), tdist.get_world_size(group)

    def _poll(self, work, what, deadline_s=None):
        limit = self._deadline(deadline_s)
        t0 = time.monotonic()
        while not work.is_completed():
            el = time.monotonic() - t0
            if limit > 0 and el > limit:
                raise CommTimeout(f"rank {self.rank} of {self.world}: {what or 'collective'} not complete after {el:.1f} s "
                                  f"(deadline {limit:g} s): a peer is dead or is not issuing the same collectives")
            time.sleep(min(1e-3, 1e-5 + el / 20))
        work.wait()                                          # complete: surfaces the collective's own exception, if any

    def _run(self, work, what, async_op):
        if async_op:
            return _PolledWork(self, work, what)
        self._poll(work, what)
        return _Done()

    def host_wait(self, stream=None, what="", deadline_s=None):
        return True                                          # collectives of this communicator complete on the host

    def all_reduce(self, t, op="sum", async_op=False):
        ops = {"sum": self.tdist.ReduceOp.SUM, "max": self.tdist.ReduceOp.MAX, "min": self.tdist.ReduceOp.MIN}
        return self._run(self.tdist.all_reduce(t, op=ops[op], group=self.group, async_op=True), "all_reduce", async_op)

    def all_gather_into(self, out, inp, async_op=False):
        return self._run(self.tdist.all_gather_into_tensor(out, inp, group=self.group, async_op=True), "all_gather", async_op)

    def broadcast(self, t, src=0):
        self._run(self.tdist.broadcast(t, src=src, group=self.group, async_op=True), "broadcast", False)

    def barrier(self):
        self._run(self.tdist.barrier(group=self.group, async_op=True), "barrier", False)


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
        """Every rank's current stream has reached this point: a one-word all-reduce, then the host waits for its stream
        (deadline-bounded: host_wait)."""
        self.all_reduce(self._bar, "sum")
        self.host_wait(what="barrier")

    def host_wait(self, stream=None, what="", deadline_s=None):
        """Block the calling thread until the work enqueued so far on `stream` (default: the current stream of the
        communicator's device) is complete - the deadline-bounded form of stream.synchronize() for the multi-rank path.  Polls an
        event recorded behind that work (hipEventQuery) and, every few polls, the communicator's asynchronous error state.  At the
        deadline: dldkd_comm_abort (RCCL's device kernels poll the abort flag, so the stream drains), then CommTimeout."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("comm.host_wait under a graph capture: the host cannot wait for work that is not running")
        ev = torch.cuda.Event()
        ev.record(s)
        return self._wait_event(ev, what or "stream work", deadline_s)

    def _wait_event(self, ev, what, deadline_s):
        limit = self._deadline(deadline_s)
        t0, n = time.monotonic(), 0
        while not ev.query():
            n += 1
            el = time.monotonic() - t0
            if n % 64 == 0 or el > 0.05:
                try:
                    self.check_async()
                except Exception:
                    self.abort()
                    raise
            if limit > 0 and el > limit:
                self.abort()
                raise CommTimeout(f"rank {self.rank} of {self.world}: {what} not complete after {el:.1f} s "
                                  f"(deadline {limit:g} s, DLDKD_COMM_DEADLINE_S): a peer is dead or is not issuing the same "
                                  "collectives; communicator aborted")
            if el > 0.005:                                   # short waits (a 2-ms step) spin; long ones leave the core alone
                time.sleep(min(2e-3, el / 50))
        self.check_async()
        return True

    def watch(self, stream=None, what=""):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        prev = getattr(self, "_watch_ev", None)
        if prev is not None and not prev.query():
            self._wait_event(prev, what or "watched work", None)
        self.check_async()
        self._watch_ev = torch.cuda.Event()
        self._watch_ev.record(s)
        return True

    def abort(self):
        """Tear the communicator down without waiting for its outstanding collectives (dldkd_comm_abort)."""
        if self._h:
            h, self._h = self._h, ctypes.c_void_p()
            try:
                self.native.check(self.lib.dldkd_comm_abort(h), "comm_abort")
            except Exception:   # noqa: BLE001 - the abort is best effort: the caller is already raising
                pass

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
