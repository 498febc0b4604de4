"""Data-parallel gradient-bucket logic on CPU: 2 processes, gloo backend.

The HIP kernels cannot run here, so the test drives exactly the host-side pieces that the
multi-GPU path adds on top of the single-GPU backward: the per-layer flat gradient buckets
(views handed to the kernels), the bucket hook that averages a bucket across ranks as soon as
it is complete, and the flat reduction of the few parameters outside the encoder Function.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, bpe=3, coll=None):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.manual_seed(rank)          # ranks are seeded DIFFERENTLY: construction must broadcast rank 0's weights
        from tim_amd.dp import DataParallel, allreduce_buckets_reference
        from tim_amd.tim import TIM, _GradBuckets
        cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
        m = TIM(cfg.num_class, visual_input_dim=24, audio_input_dim=40, d_model=32, nhead=2, num_layers=2, num_feats=6)
        before = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
        dp = DataParallel(m, buckets_per_exchange=bpe, collective=coll)
        assert dp.collective == (coll or "a2a"), (dp.collective, dp._why)
        assert dp.wire_dtype == torch.float32          # the default is the reference's exact fp32 mean
        assert dp.world == world and dp.active and m.rt.bucket_hook is not None
        names = m._encoder_param_names
        params = m._encoder_param_list()
        # every rank now holds rank 0's parameters
        flat0 = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
        ref = flat0.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, flat0)
        assert rank == 0 or not torch.equal(before, flat0)
        # ... buckets: one per layer + front + heads, views alias the flat buffers
        gb = _GradBuckets(m.rt, names, params, torch.device("cpu"), m._bucket_of)
        assert set(gb.flat) == {"front", "heads", "layer0", "layer1"}
        for i, n in enumerate(names):
            gb.views[n].fill_(float(i + 1) * (rank + 1))  # what the kernels would have accumulated
        for b in ("heads", "layer1", "layer0", "front"):  # order in which the backward completes them
            gb.done(b)
        assert len(dp._pending) == 4 % bpe                # bpe buckets travel as one contiguous range, the rest is waiting
        m.rt.finish_hook()                                # end of the encoder backward: the rest is exchanged
        assert not dp._pending
        mean = (1 + world) / 2.0
        for i, n in enumerate(names):
            v = gb.views[n]
            assert v.shape == dict(zip(names, params))[n].shape
            assert torch.allclose(v, torch.full_like(v, (i + 1) * mean)), n
        # the few parameters outside the encoder Function (time MLP, DRLoc MLP): one flat exchange
        for j, p in enumerate(dp._small):
            p.grad = torch.full_like(p, float(j + 1) * (rank + 1))
        dp._reduce_small()
        for j, p in enumerate(dp._small):
            assert torch.allclose(p.grad, torch.full_like(p, (j + 1) * mean))
        assert len(dp._small) == 8 + 6
        # bf16 on the wire, fp32 accumulation: random gradients of an odd length against the exact fp32 mean
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(100003, generator=g) * 1e-3
        exact = x.clone()
        allreduce_buckets_reference([exact], world)
        dp.wire_dtype = torch.bfloat16 if coll is None else torch.float32   # (fp32 accumulation of a 16-bit wire: the a2a form only)
        dp._stage.clear()
        got = x.clone()
        dp._exchange(got)
        # two roundings to bf16 (each rank's contribution, then the mean): 2^-8 relative to the larger of the two
        if coll is None:
            assert (got - exact).abs().max().item() <= 2.0 ** -8 * x.abs().max().item() * 1.01
            assert (got - exact).abs().mean().item() <= 2.0 ** -9 * x.abs().mean().item()
        else:
            assert torch.allclose(got, exact, rtol=1e-6, atol=1e-9)
        both = got.clone()
        dist.broadcast(both, 0)
        assert torch.equal(both, got)          # every rank ends with the SAME gradients (all-gather of the reduced shards)
        assert dp.bytes_on_wire > 0
        # the plain fp32 all-reduce path (what the group falls back to TOGETHER, test_collective_choice_is_collective)
        assert dp.collective == (coll or "a2a")
        dp.collective = "allreduce"       # (every rank of this test alike)
        z = x.clone()
        dp._exchange(z)
        assert torch.allclose(z, exact, rtol=1e-6, atol=1e-9)
        dp.collective = coll or "a2a"
        # odd lengths that W does not divide, both wire formats, against the exact mean (staging path with padded chunks)
        for n_odd in (1, 7, 100003, 64 * 5):
            for wd in ((torch.float32, torch.bfloat16) if coll is None else (torch.float32,)):
                dp.wire_dtype = wd
                dp._stage.clear()
                xo = torch.randn(n_odd, generator=g) * 1e-3
                ex = xo.clone()
                allreduce_buckets_reference([ex], world)
                go = xo.clone()
                dp._exchange(go)
                mx = xo.abs().max().clone()
                dist.all_reduce(mx, op=dist.ReduceOp.MAX)          # the largest contribution of any rank sets the rounding step
                lim = 1e-9 + (2.0 ** -8 * 1.01 if wd == torch.bfloat16 else 1e-6) * mx.item()
                assert (go - ex).abs().max().item() <= lim, (n_odd, wd)
        dp.wire_dtype = torch.float32
        dp._stage.clear()
        # no_sync(): gradients stay local
        with dp.no_sync():
            y = torch.full((64,), float(rank + 1))
            dp._on_bucket("layer0", y)
            assert torch.equal(y, torch.full((64,), float(rank + 1)))
        dp._accumulating = False
        # gradient accumulation: microbatch 1 under no_sync(), microbatch 2 synchronised -> p.grad = mean over ranks of (g1 + g2)
        # for EVERY parameter.  The kernels cannot run here; the test plays autograd's part (AccumulateGrad assigns the returned
        # view when p.grad is None, adds to it otherwise) around the real bucket objects and hooks.
        pd = dict(zip(names, params))
        for p in m.parameters():
            p.grad = None

        def one_pass(scale):
            gbk = _GradBuckets(m.rt, names, params, torch.device("cpu"), m._bucket_of)
            for i, n in enumerate(names):
                gbk.views[n].fill_(scale * float(i + 1) * (rank + 1))
            for b in ("heads", "layer1", "layer0", "front"):
                gbk.done(b)
            m.rt.finish_hook()
            for n in names:                                   # AccumulateGrad
                if pd[n].grad is None:
                    pd[n].grad = gbk.views[n]
                else:
                    pd[n].grad += gbk.views[n]
            for j, p in enumerate(dp._small):
                gs = torch.full_like(p, scale * float(j + 1) * (rank + 1))
                p.grad = gs if p.grad is None else p.grad + gs
            dp._reduce_small()

        with dp.no_sync():
            one_pass(1.0)
        for i, n in enumerate(names):
            assert torch.allclose(pd[n].grad, torch.full_like(pd[n], float(i + 1) * (rank + 1)))   # still local
        one_pass(10.0)
        for i, n in enumerate(names):
            assert torch.allclose(pd[n].grad, torch.full_like(pd[n], 11.0 * (i + 1) * mean)), n
        for j, p in enumerate(dp._small):
            assert torch.allclose(p.grad, torch.full_like(p, 11.0 * (j + 1) * mean))
        assert not dp._accumulating
        one_pass(100.0)   # a further synchronised pass without zero_grad: plain accumulation of averaged gradients
        for i, n in enumerate(names):
            assert torch.allclose(pd[n].grad, torch.full_like(pd[n], 111.0 * (i + 1) * mean)), n
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


def _choice_worker(rank, world, port, q, mode):
    """mode "preflight": rank 1 alone refuses the all-to-all path before any collective; "env": rank 0 alone carries the A/B
    switch; "raise": the all-to-all raises (on every rank: a backend without it); "wrong": rank 1's probe comes back with
    a wrong mean; "step_error": after a healthy construction a step's exchange raises on one rank - and must propagate."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import warnings
        from tim_amd import dp as DP
        from tim_amd.tim import TIM
        cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
        m = TIM(cfg.num_class, visual_input_dim=24, audio_input_dim=40, d_model=32, nhead=2, num_layers=2, num_feats=6)
        a2a_calls = []
        real_a2a = dist.all_to_all_single
        if mode == "preflight" and rank == 1:
            def refuse(self):
                raise RuntimeError("forced: this rank cannot run the all-to-all path")
            DP.DataParallel._preflight = refuse
        if mode == "env" and rank == 0:
            os.environ["TIM_AMD_DP_COLLECTIVE"] = "allreduce"
        if mode == "invalid_env" and rank == 0:
            # an invalid value on rank 0 alone: a refusal like any other (the whole group lands on all_reduce); rank 0 must not
            # raise out of the constructor on its own afterwards, with its peers inside broadcast_parameters()
            os.environ["TIM_AMD_DP_COLLECTIVE"] = "ring_of_fire"
        if mode == "raise":
            def broken(*a, **k):
                raise RuntimeError("forced: backend has no all_to_all")
            DP.dist.all_to_all_single = broken
        if mode == "wrong" and rank == 1:
            orig = DP.DataParallel._reduce_chunks

            def off_by_one(self, st, W, per, on_gpu):
                orig(self, st, W, per, on_gpu)
                st["shard"].add_(1.0)
            DP.DataParallel._reduce_chunks = off_by_one
        if mode in ("preflight", "env"):
            def counted(*a, **k):
                a2a_calls.append(1)
                return real_a2a(*a, **k)
            DP.dist.all_to_all_single = counted
        with warnings.catch_warnings(record=True) as wrn:
            warnings.simplefilter("always")
            # "rs_on_gloo": the reduce-scatter + all-gather form (the default over RCCL) asked for explicitly - gloo carries it too;
            # "mixed": the ranks ask for different forms and land on all_reduce together
            kw = {"collective": "rs_ag"} if mode == "rs_on_gloo" else ({"collective": ("a2a", "rs_ag")[rank]} if mode == "mixed" else {})
            dp = DP.DataParallel(m, **kw)
        want = "a2a" if mode == "step_error" else ("rs_ag" if mode == "rs_on_gloo" else "allreduce")
        assert dp.collective == want, dp.collective
        agreed = torch.tensor([1.0 if dp.collective == "allreduce" else 0.0])
        lo, hi = agreed.clone(), agreed.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert lo.item() == hi.item()                      # THE SAME decision on every rank
        if mode in ("preflight", "env"):
            assert not a2a_calls                           # nobody issued an all-to-all a peer would not join
        if mode not in ("env", "step_error", "rs_on_gloo") and rank == 0:
            assert any("agreed on plain all_reduce" in str(w.message) for w in wrn)
        if mode == "wrong" and rank == 1:
            DP.DataParallel._reduce_chunks = orig
        if mode == "step_error":
            # no per-rank fallback any more: an error inside a step's exchange propagates on the rank that saw it
            if rank == 1:
                def boom(*a, **k):
                    raise RuntimeError("forced: out of memory")
                dp._exchange_a2a = boom
                try:
                    dp._exchange(torch.ones(64))
                    raise AssertionError("the error was swallowed")
                except RuntimeError as e:
                    assert "forced" in str(e)
                assert dp.collective == "a2a"
        else:
            # the fallback exchanges correctly, on every rank
            x = torch.full((1000,), float(rank + 1))
            dp._exchange(x)
            assert torch.allclose(x, torch.full((1000,), (1 + world) / 2.0))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("mode", ["preflight", "env", "invalid_env", "raise", "wrong", "step_error", "rs_on_gloo", "mixed"])
def test_collective_choice_is_collective(mode):
    """tim_amd/dp.py:_choose_collective - the all-to-all -> all-reduce fallback is decided once, by the whole group: a refusal,
    an exception or a wrong probe result on ONE rank moves EVERY rank to all_reduce before a step runs; after construction
    nothing switches collectives on its own (an error in a step's exchange is raised, not absorbed)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_choice_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


@pytest.mark.parametrize("world,bpe,coll", [(2, 3, None), (3, 1, None), (4, 4, None), (2, 2, "rs_ag"), (3, 4, "rs_ag")])
def test_gradient_buckets_gloo(world, bpe, coll):
    """world sizes 2, 3 (divides no bucket: padded chunks) and 4; one bucket per exchange, three, and all four as one range; the
    all-to-all form (gloo's default here) and the reduce-scatter + all-gather form (the default over RCCL)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, bpe, coll)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_bench_shards_windows_across_ranks():
    """bench.py gives every rank its own windows (weak scaling, no data-path collective)."""
    import numpy as np
    from tim_amd import synth
    from tim_amd.config import named_config
    cfg = named_config("tiny")
    a = synth.make_inputs(cfg, 2, 4, 2, seed=100)
    b = synth.make_inputs(cfg, 2, 4, 2, seed=101)
    assert not np.allclose(a["visual"], b["visual"])
    assert np.allclose(a["times"][:, :12], b["times"][:, :12])  # feature times are the window grid
