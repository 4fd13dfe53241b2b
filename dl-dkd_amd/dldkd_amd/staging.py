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
