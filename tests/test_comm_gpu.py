"""GPU, one rank: the RCCL communicator of the C ABI (dldkd_comm_*, dldkd_amd.comm.RcclComm) - the collectives of the sharded
evaluation (method/eval.py:188-212 cut by video) and of the data-parallel step (method/train.py:147-151).  With one rank a sum /
max / min over ranks and a gather are identities and RCCL still goes through its enqueue path (stream order, buffers, dtypes,
counts): results must be bit-identical to the inputs.  What one GPU cannot show - the arithmetic over several ranks - is covered
over gloo at world size 2 (tests/test_dist_cpu.py) through the same dist.py code."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_collectives_one_rank_bit_identical(rccl_comm):
    from dldkd_amd import native
    c = rccl_comm
    assert (c.rank, c.world) == (0, 1)
    w, r = ctypes.c_int(-1), ctypes.c_int(-1)
    native.check(native.lib().dldkd_comm_info(c._h, ctypes.byref(w), ctypes.byref(r)), "comm_info")
    assert (w.value, r.value) == (1, 0)
    assert native.lib().dldkd_comm_rccl_version() >= 21000
    g = torch.Generator(device=DEV).manual_seed(1)
    for dtype in (torch.float32, torch.float64, torch.int32, torch.int64, torch.uint8):
        for n in (1, 257, 4_371_968 if dtype == torch.float32 else 1000):       # 17.5 MB: the Charades gradient buffer
            x = (torch.randn(n, generator=g, device=DEV) * 100).to(dtype)
            for op in ("sum", "max", "min"):
                y = x.clone()
                c.all_reduce(y, op)
                assert torch.equal(y, x), (dtype, n, op)
            out = torch.empty(n, dtype=dtype, device=DEV)
            c.all_gather_into(out, x)
            assert torch.equal(out, x)
            b = x.clone()
            c.broadcast(b, 0)
            assert torch.equal(b, x)
    c.barrier()
    c.check_async()
    assert c.max_over_ranks(3.25, torch.device(DEV)) == 3.25
    # argument errors are RuntimeErrors with a text, raised before anything is enqueued
    with pytest.raises(native.NativeError):
        c.all_reduce(torch.zeros(4), "sum")                                    # CPU tensor
    with pytest.raises(native.NativeError):
        c.all_reduce(torch.zeros(4, 4, device=DEV).t(), "sum")                  # not contiguous
    with pytest.raises(native.NativeError):
        c.all_gather_into(torch.zeros(5, device=DEV), torch.zeros(4, device=DEV))
    with pytest.raises(native.NativeError):
        c.all_reduce(torch.zeros(4, device=DEV, dtype=torch.float16), "sum")
    L = native.lib()
    assert L.dldkd_comm_all_reduce(None, None, None, 4, 0, 0, None) == -1 and b"dldkd_comm_all_reduce" in L.dldkd_last_error()
    assert L.dldkd_comm_all_reduce(c._h, None, None, 0, 0, 0, None) == 0      # nothing to do is not an error
    assert L.dldkd_comm_all_reduce(c._h, ctypes.c_void_p(8), ctypes.c_void_p(8), 4, 99, 0, None) == -1


def test_collectives_are_plain_stream_work(rccl_comm):
    """A collective is ordered by the stream it is enqueued on like any launch: issued on a side stream behind an event it sees
    the producer's data; between two hipGraph replays it sees the first replay's output; captured INTO a graph it is replayed
    with it.  No thread of the process polls anything (the r04 process-group watchdog did, and aborted beside captures)."""
    c = rccl_comm
    side = torch.cuda.Stream(device=DEV)
    x = torch.zeros(1 << 20, device=DEV)
    out = torch.empty_like(x)
    for i in range(20):
        x.fill_(float(i))
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            c.all_gather_into(out, x)
            c.all_reduce(out, "sum")
        torch.cuda.current_stream().wait_stream(side)
        assert float(out[12345]) == float(i) and float(out.min()) == float(i)
    # between graph replays, on the graphs' stream
    st = torch.cuda.Stream(device=DEV)
    a = torch.zeros(4096, device=DEV)
    b = torch.zeros(4096, device=DEV)
    with torch.cuda.stream(st):
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        a.add_(1.0)                                                            # (warm-up outside the captures)
        with torch.cuda.graph(g1, stream=st, capture_error_mode="thread_local"):
            a.add_(1.0)
        with torch.cuda.graph(g2, stream=st, capture_error_mode="thread_local"):
            b.copy_(a * 2.0)
        a.zero_()
        for i in range(10):
            g1.replay()
            c.all_reduce(a, "sum")
            g2.replay()
        st.synchronize()
    assert float(a[0]) == 10.0 and float(b[7]) == 20.0
    # inside a capture: the gather becomes a node of the graph
    src = torch.arange(1024, device=DEV, dtype=torch.float32)
    dst = torch.zeros(1024, device=DEV)
    with torch.cuda.stream(st):
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, stream=st, capture_error_mode="thread_local"):
            c.all_gather_into(dst, src)
            dst.mul_(2.0)
        for k in (1.0, 3.0):
            src.fill_(k)
            g3.replay()
            st.synchronize()
            assert float(dst[5]) == 2.0 * k
    c.check_async()


def test_communicators_come_and_go():
    """Create / use / destroy, several times in one process (a trainer that evaluates with a fresh communicator per phase, the test
    suite itself): no state survives in the library."""
    from dldkd_amd import comm as dcomm
    x = torch.ones(1000, device=DEV)
    for i in range(4):
        c = dcomm.RcclComm(1, 0, dcomm.RcclComm.unique_id(), torch.device(DEV))
        c.all_reduce(x, "sum")
        c.barrier()
        c.destroy()
        c.destroy()                                                            # idempotent
    assert float(x.sum()) == 1000.0
    assert dcomm.current() is None and dcomm.info() == (0, 1)


@pytest.mark.parametrize("in_graph", [False, True])
def test_data_parallel_replay_uses_the_tower_graphs_with_one_all_reduce(rccl_comm, in_graph):
    """The data-parallel step of the default layout (one gradient bucket): the same tower graphs as on one GPU, ONE all-reduce
    enqueued between the backward graphs and the optimizer graph, the optimizer graph computing the clip's norms from the reduced
    gradients.  Against the one-GPU stepper from the same state on the same seeds: same loss, same parameters (a mean over one rank
    is the identity), step after step, with dropout."""
    import types
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import synth
    from dldkd_amd import train as T
    from dldkd_amd.model import DLDKD
    from dldkd_amd.optimization import BertAdam
    cfg = types.SimpleNamespace(visual_input_size=256, query_input_size=128, inheritance_hidden=384, exploration_hidden=384,
                                max_ctx_l=32, max_desc_l=30, input_drop=0.2, drop=0.2, n_heads=4, initializer_range=0.02,
                                margin=0.1, use_hard_negative=True, hard_pool_size=5, label_style="soft")
    mopt = types.SimpleNamespace(double_branch=True, kl_intra_weight=0.1, inher_nce_weight=0.04, explore_nce_weight=0.04,
                                 collection="tvr", alpha=0.8, belta=0.8)
    topt = types.SimpleNamespace(grad_clip=-1)
    batches = [synth.make_train_batch(270 + i, nv=24, caps=2, L=32, len_lo=3, dv=256, dq=128, lq_lo=6, lq_hi=30) for i in range(2)]
    batches = [{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]

    def make():
        torch.manual_seed(11)
        m = DLDKD(types.SimpleNamespace(**vars(cfg)), mopt).to(DEV).train()
        return m, BertAdam([{"params": list(m.parameters()), "weight_decay": 0.01}], lr=2e-3, warmup=0.1, t_total=40)

    old = T.DDP_MIN_WORLD
    try:
        T.DDP_MIN_WORLD = 2
        mp_, op_ = make()
        plain = T.GraphedTrainStep(mp_, op_, topt)
        T.DDP_MIN_WORLD = 1
        md, od = make()
        # (as under a real data-parallel run, world >= 2: no self-check - it would step the capture batch twice, with two all-reduces)
        ddp = T.GraphedTrainStep(md, od, types.SimpleNamespace(grad_clip=-1, graph_self_check=False, ddp_allreduce_in_graph=in_graph))
        assert ddp.self_check is False and plain.self_check is True
        calls = []
        real = rccl_comm.all_reduce
        rccl_comm.all_reduce = lambda t, op="sum", async_op=False: (calls.append((t.numel(), op)), real(t, op, async_op))[1]
        from dldkd_amd import ops
        replays_before = captures_before = 0
        for it in range(8):
            # both steppers start every step from the same state (the step's fp32 atomics make two runs drift apart at rounding
            # level per step; the test is about one step's arithmetic, step after step)
            od.fp.flat.copy_(op_.fp.flat); od.m.copy_(op_.m); od.v.copy_(op_.v)
            od.step_count = op_.step_count
            ops.bump_param_epoch()
            T.DDP_MIN_WORLD = 2
            torch.manual_seed(500 + it)
            lp, _ = plain(batches[it % 2])
            T.DDP_MIN_WORLD = 1
            torch.manual_seed(500 + it)
            n0 = len(calls)
            ld, _ = ddp(batches[it % 2])
            grads = [c_ for c_ in calls[n0:] if c_[0] == od.fp.grad.numel()]
            # exactly one gradient all-reduce per step: enqueued between the graphs, or (in_graph) captured once into the optimizer
            # graph - the capture step sees the call, the replays run the node
            replayed = ddp.replays > replays_before
            replays_before = ddp.replays
            expect = 0 if (in_graph and replayed and ddp.captures == captures_before) else 1
            captures_before = ddp.captures
            assert len(grads) == expect and all(g_[1] == "sum" for g_ in grads), (it, in_graph, calls[n0:])
            assert float(lp) == pytest.approx(float(ld), rel=1e-5), it
            tol = 2e-7 + 0.02 * od.get_lr()[0]
            assert (op_.fp.flat - od.fp.flat).abs().max().item() <= tol, it
        rccl_comm.all_reduce = real
        assert ddp.captures >= 1 and ddp.replays >= 5 and ddp.capture_failures == 0 and ddp.fallbacks == []
        e = next(iter(ddp.graphs.values()))
        assert e.ddp and getattr(e, "par", None), "the data-parallel step should replay the tower graphs"
        assert e.par["ar_in_graph"] is in_graph
    finally:
        T.DDP_MIN_WORLD = old


def test_host_wait_is_deadline_bounded_and_aborts(rccl_comm):
    """the blocking points of the multi-rank path poll an event + the communicator's error state against a
    deadline instead of synchronising for ever.  A stream busy for ~4 s (a spin kernel: it ENDS by itself, so nothing here can
    hang the box - ncclCommAbort frees device memory, which waits for the device) stands in for a collective whose peer is late:
    host_wait raises CommTimeout at its 1-s deadline and leaves the communicator aborted; work that does finish returns at once;
    watch() leaves markers without draining the stream."""
    import time
    from dldkd_amd import comm as dcomm
    from dldkd_amd import native
    c = rccl_comm
    x = torch.ones(1 << 20, device=DEV)
    c.all_reduce(x, "sum")
    t0 = time.monotonic()
    assert c.host_wait(what="a finished all-reduce") is True and time.monotonic() - t0 < 5.0
    assert c.watch() and c.watch()                                   # second call checks the first marker: complete
    # calibrate the spin kernel, then keep a side stream busy for ~4 s
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
    per_ms = 20_000_000 / max(e0.elapsed_time(e1), 1e-3)
    busy = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(busy):
        for _ in range(8):
            torch.cuda._sleep(int(per_ms * 500))                     # 8 x 0.5 s
        y = x * 2
    t0 = time.monotonic()
    with pytest.raises(dcomm.CommTimeout) as ei:
        c.host_wait(busy, what="a stream that is late", deadline_s=1.0)
    dt = time.monotonic() - t0
    assert 0.9 <= dt < 30.0 and "deadline 1 s" in str(ei.value) and "aborted" in str(ei.value)
    assert not c._h                                                  # aborted: the handle is gone ...
    with pytest.raises(native.NativeError):
        c.all_reduce(x, "sum")                                       # ... and later collectives fail loudly instead of hanging
    busy.synchronize()
    assert float(y[0]) == 2.0
