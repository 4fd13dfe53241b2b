"""Per-step host scalars -> device without blocking the host and without races.

A training step needs a few host-produced values on the device: the learning rates of the step, the reference's CPU
torch.randint draws for the triplet negatives (model.py:366-380), the batch's labels, the Philox (seed, offset) of the dropout
masks.  A pageable `tensor.to(device)` blocks the host until the stream has drained; an asynchronous copy from ONE pinned
buffer races with the host, which runs several steps ahead of the GPU and would overwrite the buffer before the copy engine has
read it.  PinnedRing hands out pinned slots round robin and remembers, per slot, an event recorded after the slot's copy was
enqueued: a slot is only handed out again once that copy has executed (normally long ago, so the wait is free)."""
import time

import torch


class PinnedRing:
    def __init__(self, nbytes, device, slots=4):
        self.cuda = torch.device(device).type == "cuda"
        self.bufs = [torch.zeros(max(nbytes, 16), dtype=torch.uint8, pin_memory=self.cuda) for _ in range(slots)]
        self.events = [None] * slots
        self.i = -1

    def next(self):
        """The next free pinned slot (uint8 tensor); blocks only if its previous upload has not executed yet."""
        self.i = (self.i + 1) % len(self.bufs)
        ev = self.events[self.i]
        if ev is not None and not ev.query():
            # The host is a whole ring ahead of the GPU.  Poll before parking: hipEventSynchronize on an incomplete event was
            # measured at 0.5-2.4 ms per call on a loaded host (bench.py after its CPU baseline), i.e. the GPU had long drained
            # its queue when the host woke up - the gallery encode ran at a quarter of its speed in such runs.
            t_end = time.perf_counter() + 2e-3
            while not ev.query():
                if time.perf_counter() > t_end:
                    ev.synchronize()
                    break
        return self.bufs[self.i]

    def upload(self, dev_bytes, by_kernel=False):
        """Enqueue the asynchronous copy of the current slot into `dev_bytes` (uint8 device tensor of the same size) on the
        current stream and remember when it is done.  by_kernel: a few KB (size a multiple of 4) read from the pinned slot by a
        kernel on the compute queue instead of the copy engine (native dldkd_upload_words says why)."""
        if by_kernel and self.cuda:
            from . import native
            native.check(native.lib().dldkd_upload_words(self.bufs[self.i].data_ptr(), native.ptr(dev_bytes), dev_bytes.numel() // 4,
                                                         native.stream()), "upload_words")
        else:
            dev_bytes.copy_(self.bufs[self.i][:dev_bytes.numel()], non_blocking=True)
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.events[self.i] = ev

    def upload_range(self, dev_bytes, lo, hi, last):
        """Bytes [lo, hi) of the current slot into the same range of `dev_bytes` (asynchronous, current stream): a slot uploaded in
        pieces - what the first consumers need right away, the rest once the host has produced it.  last: this piece completes
        the slot (its event is recorded here)."""
        dev_bytes[lo:hi].copy_(self.bufs[self.i][lo:hi], non_blocking=True)
        if last and self.cuda:
            ev = torch.cuda.Event()
            ev.record()
            self.events[self.i] = ev


def concurrent_streams(device, n, candidates=16, cycles=1_000_000, high_priority=0):
    """n torch streams that run side by side on the device.  HIP multiplexes its streams onto GPU_MAX_HW_QUEUES (default 4)
    hardware queues, and two streams that share a queue execute their kernels in order: of torch's pooled streams number 0 shares
    a queue with 7 and 11, 3 with 4 and 8, ... (tools/probe_stream_queues.py) - the training stepper's four tower streams were
    two pairs on two queues.  The mapping is not visible through the API, so it is measured once: a one-workgroup spin kernel
    (torch.cuda._sleep) on a pair of streams takes one kernel time when they have their own queues and two when they share
    one.  Greedy: a candidate joins the set when it overlaps with every stream already in it."""
    import time
    device = torch.device(device)
    # high_priority: the first that many streams of the result are high-priority streams (their kernels are dispatched ahead of the
    # others' when both are ready: the short query-tower chains of the training step next to the chip-filling video towers)
    cands = [torch.cuda.Stream(device=device, priority=-1) for _ in range(max(candidates // 2, high_priority) if high_priority else 0)]
    n_high = len(cands)
    cands += [torch.cuda.Stream(device=device) for _ in range(max(candidates, n))]

    def run(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    run(cands[:1])
    one = min(run(cands[:1]) for _ in range(3))
    chosen = [cands[0]]
    for j, c in enumerate(cands[1:], 1):
        if len(chosen) == n:
            break
        if j < n_high and len(chosen) >= high_priority:      # enough high-priority streams: on to the normal ones
            continue
        if all(min(run([c, o]) for _ in range(2)) < 1.5 * one for o in chosen):
            chosen.append(c)
    for c in cands:                    # fewer than n distinct queues (GPU_MAX_HW_QUEUES < n): fill up with what there is
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    return chosen


_MEMSET_PROBE = {}


def memset_node_defect(device, replays=48, log=None, select=False):
    """True / False: does a replayed hipGraph MEMSET node leave stale words on this runtime?  Probed once per process and device:
    the native call that zeroes a scratch buffer by hipMemsetAsync is captured as a one-node graph over 74 floats (BertAdam's
    296-byte norm scratch, where ROCm 7.0.2 showed the defect) and replayed on a drained stream over a freshly poisoned buffer;
    the graph is [memset node, a kernel that adds 1 to every word] - the optimizer's [zero the norms, accumulate into them] - and a
    word that is not exactly 1.0 afterwards is the defect.  select=True (opt.scratch_zeroing = "probe"): the native zeroing mode
    follows the result - clean -> memset nodes, defect or a probe that could not run -> kernel fills.  The default keeps the
    kernel fills whatever the probe says and only logs it: on the round-4 boxes (same ROCm 7.0.2) this probe found 0 stale words in
    48 replays although round 3 saw the defect inside the full optimizer graph - a clean probe is weak evidence, a dirty one is
    proof."""
    device = torch.device(device)
    if device.type != "cuda":
        return False
    if device not in _MEMSET_PROBE:
        from . import native
        lib = native.lib()
        stale, err = 0, None
        prev = lib.dldkd_set_zero_by_memset(1)
        try:
            st = torch.cuda.Stream(device=device)
            buf = torch.empty(74, dtype=torch.float32, device=device)
            ones = torch.ones(74, dtype=torch.float32, device=device)
            torch.cuda.synchronize(device)
            with torch.cuda.stream(st):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                    native.check(lib.dldkd_zero_scratch_f32(native.ptr(buf), 74, native.stream()), "zero_scratch")
                    native.check(lib.dldkd_axpy_f32(native.ptr(buf), native.ptr(ones), 1.0, 74, native.stream()), "axpy")
                for r in range(int(replays)):
                    buf.view(torch.int32).fill_(0x510c7186 + r)
                    st.synchronize()                       # the graph is launched on an idle stream
                    g.replay()
                    st.synchronize()
                    stale += int((buf != 1.0).sum())
            del g
        except Exception as ex:                            # noqa: BLE001 - a probe that cannot run selects the safe mode
            err = repr(ex)
        finally:
            lib.dldkd_set_zero_by_memset(prev)
        defect = err is not None or stale != 0
        _MEMSET_PROBE[device] = (defect, stale, err)
        if log is not None:
            log(f"hipGraph memset-node probe on {device}: {stale} stale words in {replays} replays" + (f" (probe failed: {err})" if err else ""))
    defect = _MEMSET_PROBE[device][0]
    if select:
        from . import native
        native.lib().dldkd_set_zero_by_memset(0 if defect else 1)
    return defect
