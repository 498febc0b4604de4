"""Data-parallel wrapper: one process per GPU, gradient mean over RCCL (torch.distributed
backend "nccl" on ROCm) — the replacement for the DistributedDataParallel wrap of
recognition/time_interval_machine/models/build.py:58-63.

Windows are independent, so the only exchange per step is the gradient mean.  The encoder
backward produces one flat fp32 bucket per layer (plus a heads and a front-end bucket); each
bucket is exchanged on a side stream the moment its layer's backward has been enqueued, so the
exchange of layer l overlaps the backward of layers l-1..0.

The exchange is SURVEY 8e's pattern - reduce-scatter, then all-gather - in one of three forms the
whole group agrees on at construction (`collective`, `_choose_collective`):

    "rs_ag"      dist.reduce_scatter_tensor + dist.all_gather_into_tensor on the fp32 bucket itself: RCCL
                 sums the W contributions of chunk r in fp32 on their way to rank r, the W means are
                 gathered back into the bucket.  No receive buffer, no local sum pass.  THE DEFAULT for an
                 fp32 wire over RCCL, and the form a HIP graph can carry: the data-parallel step - its
                 collectives on the comm stream included - is captured by tim_amd.graph.GraphedStep and
                 replayed (tested on a one-rank RCCL group: tests/test_gpu_dp.py, tools/dp_graph_check.py).
    "a2a"        all-to-all (rank r receives chunk r of every rank's bucket), local fp32 sum
                 (timhip_dp_reduce), all-gather.  xGMI is a full mesh of point-to-point links (7 x ~153 GB/s
                 per GPU), an all-to-all drives all seven at once whatever ring RCCL would build; and it is
                 the only form that accumulates a 16-bit wire in fp32 (`wire_dtype=torch.bfloat16`: 117
                 instead of 233 MB per step and direction for C2a's 58.3 M parameters, each contribution and
                 the mean rounded to 8 mantissa bits - a change in training numerics the caller has to
                 want).  Default for 16-bit wires and for backends other than RCCL.  EAGER ONLY:
                 all_to_all_single is send / receive pairs underneath, and captured they hang or crash
                 hipStreamEndCapture on this stack (ROCm 7.0, RCCL 2.26: profiles/r05_rccl_capture_probe.txt).
    "allreduce"  one fp32 all-reduce per range: what the group falls back to TOGETHER when a rank refuses
                 the preferred form or the probe exchange fails anywhere.

`wire_dtype=torch.float32` (default) gives the exact fp32 mean, what the reference's
DistributedDataParallel computes.  Which of "rs_ag" / "a2a" is faster on eight real GPUs is not known:
no multi-GPU box has been available to this repo; TIM_AMD_DP_COLLECTIVE=a2a|rs_ag|
allreduce switches without a code change.  The few parameters outside the encoder Function (time MLP,
DRLoc MLP: ~1.6 M) go through the same exchange as one more bucket when the backward finishes.

Gradient accumulation (`no_sync()`): passes inside the context leave their gradients in `p.grad`,
un-exchanged.  The next synchronised pass folds those local sums into its buckets before the
exchange and hands autograd the averaged total, so after it `p.grad = mean over ranks of
(g_1 + ... + g_k)` for every parameter - the DistributedDataParallel result.

Like DistributedDataParallel, construction broadcasts rank 0's parameters and buffers, so ranks
that were seeded or loaded differently start from the same weights.
"""
import contextlib
import os

import torch
import torch.distributed as dist
from torch import nn


class DataParallel(nn.Module):
    def __init__(self, module, process_group=None, wire_dtype=torch.float32, broadcast_parameters=True, force=False,
                 buckets_per_exchange=4, collective=None):
        """collective: "rs_ag" (reduce-scatter + all-gather, RCCL sums in fp32 on the way: the default for an fp32 wire on the
        nccl / RCCL backend, and the form a HIP graph can carry), "a2a" (all-to-all + local fp32 sum + all-gather: the only form
        that accumulates a 16-bit wire in fp32; default for 16-bit wires and for backends without reduce_scatter_tensor),
        "allreduce" (one fp32 all-reduce per range), or None = the environment's TIM_AMD_DP_COLLECTIVE, else the default.  The
        group probes the preferred form together at construction and falls back to "allreduce" together.
        force: run the exchange even in a one-rank group (every collective is then a copy) - the single-GPU
        test of how the side-stream work interferes with the backward uses it.
        buckets_per_exchange: consecutive gradient buckets (contiguous in memory, tim.py:_GradBuckets) travel as one range.
        What the wrapper costs on ONE GPU before a byte crosses a link (one-rank RCCL group, every collective a copy of the
        whole range: ~0.9 GB of extra memory traffic per step beside the backward; C2a, B = 64, 5.22 ms plain step, round 5,
        profiles/r05_dp_single_gpu_overhead.txt), eager / as a graph replay:
            rs_ag      1: +0.57 / +0.51 ms   2: +0.40 / +0.46   4: +0.37 / +0.38   all: +0.48 / +0.32
            a2a        1: +0.79              2: +0.51           4: +0.55           all: +0.61          (eager only)
            allreduce  1: +0.35 / +0.38      2: +0.27 / +0.34   4: +0.22 / +0.22   all: +0.27 / +0.12  (in place on one rank)
        i.e. every exchange costs launches on the comm stream and a join, fewer and larger ranges are cheaper until nothing is
        left to overlap (eager "all": the tail is exposed; in a replay the host is out of the picture and "all" is cheapest
        on one rank - on W ranks "all" has no overlap with the backward at all, so 4 stays the default)."""
        super().__init__()
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.active = (self.world > 1 or force) and dist.is_initialized()
        self.wire_dtype = wire_dtype
        self.sync = True           # False inside no_sync(): gradients stay local
        self._comm = None
        self._callback_queued = False
        self._stage = {}           # (numel, device) -> staging buffers, reused every step
        self.comm_events = []      # (start, end) event pairs of this step's exchanges (timing=True only)
        self.timing = False
        self.bytes_on_wire = 0     # per rank and step: bytes sent + received by the last step's exchanges
        self.buckets_per_exchange = max(1, int(buckets_per_exchange))
        self._pending = []         # buckets completed but not yet exchanged: (flat, ready)
        self._accumulating = False  # a backward pass ran under no_sync() since the last exchange
        self._want = collective     # None: environment, then the default for the wire dtype / backend (_preferred)
        self.collective = "a2a"     # "rs_ag" | "a2a" | "allreduce": agreed by the whole group in _choose_collective, never changed afterwards
        self._why = ""
        self._wanted = None
        self._stub = os.environ.get("TIM_AMD_DP_STUB", "0") == "1"
        rt = module.rt
        rt.bucket_hook = self._on_bucket
        rt.finish_hook = self._on_encoder_done
        self._small = [p for n, p in module.named_parameters()
                       if n.startswith("time_mlp.") or n.startswith("drloc_mlp.") or n.startswith("pool.")]
        self._small_flat = None
        self._hook_handles = []
        if self.active:
            for p in self._small:
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._on_small_grad))
            dev = next(module.parameters()).device
            self.collective = self._choose_collective(dev)
            # (the preference is NOT re-evaluated here: an invalid collective= / TIM_AMD_DP_COLLECTIVE value raised inside
            #  _choose_collective, was counted as a refusal by the whole group, and raising now on rank 0 alone would leave
            #  the peers inside broadcast_parameters() - round-5 advisor finding)
            if self.collective == "allreduce" and self.rank == 0 and self._wanted != "allreduce":
                import warnings
                warnings.warn("tim_amd.dp: the group agreed on plain all_reduce for the gradient exchange (%s)"
                              % (self._why or "a peer refused the %s path" % self._wanted))
            if broadcast_parameters:
                self.broadcast_parameters()

    # ---- construction: everybody starts from rank 0's weights (what DDP does, build.py:58-63) ---------
    @torch.no_grad()
    def broadcast_parameters(self):
        tensors = [p.data for p in self.module.parameters()] + [b.data for b in self.module.buffers()]
        by_kind = {}
        for t in tensors:
            by_kind.setdefault((t.dtype, t.device), []).append(t)
        for (dtype, dev), ts in by_kind.items():
            flat = torch.cat([t.reshape(-1) for t in ts])      # once per run: not on the step path
            dist.broadcast(flat, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        self.module.rt.invalidate_weights()   # .data writes do not bump the version the operand-copy cache keys on

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation: backward passes inside the context leave the gradients un-exchanged (as DDP.no_sync)"""
        old, self.sync = self.sync, False
        try:
            yield
        finally:
            self.sync = old

    # ---- the exchange of one flat fp32 bucket -------------------------------------------------------
    def _staging(self, n, dev, recv_only=False, rs=False):
        """buffers of one range size, reused every step.  a2a: recv [W x per] (+ send, the padded / narrowed copy, unless the
        collectives run on the bucket itself: recv_only), shard [per], acc [per];  rs: shard [per] (+ send unless recv_only)"""
        key = (n, dev, recv_only, rs)
        st = self._stage.get(key)
        if st is None:
            W = self.world
            per = (n + W - 1) // W
            per = (per + 7) // 8 * 8                       # 16-byte chunks on the wire
            st = {"per": per,
                  "send": None if recv_only else torch.zeros(per * W, dtype=self.wire_dtype, device=dev),   # padding stays zero
                  "recv": None if rs else torch.empty(per * W, dtype=self.wire_dtype, device=dev),
                  "shard": torch.empty(per, dtype=self.wire_dtype, device=dev),
                  "acc": None if rs else torch.empty(per, dtype=torch.float32, device=dev)}
            self._stage[key] = st
        return st

    def _reduce_chunks(self, st, W, per, on_gpu):
        """shard[per] = (1/W) * sum over the W received chunks, accumulated in fp32"""
        if on_gpu and self.wire_dtype in (torch.bfloat16, torch.float32):            # one launch
            from ._lib import call, ptr
            call("timhip_dp_reduce", 1 if self.wire_dtype == torch.bfloat16 else 0, ptr(st["recv"]), W, per, 1.0 / W,
                 ptr(st["shard"]), torch.cuda.current_stream().cuda_stream)
        else:                                                                        # (gloo / CPU tests of the logic)
            torch.sum(st["recv"].view(W, per), dim=0, dtype=torch.float32, out=st["acc"])
            st["acc"].mul_(1.0 / W)
            st["shard"].copy_(st["acc"])

    # ---- which collective: decided ONCE, by every rank together ------------------------------------
    def _preferred(self):
        """the exchange form this rank would like: constructor argument, else TIM_AMD_DP_COLLECTIVE, else by wire dtype and
        backend (fp32 over RCCL: reduce-scatter + all-gather; a 16-bit wire, or a backend without reduce_scatter_tensor such as
        gloo: all-to-all + local fp32 sum + all-gather)"""
        want = self._want or os.environ.get("TIM_AMD_DP_COLLECTIVE")
        if want:
            if want not in ("rs_ag", "a2a", "allreduce"):
                raise ValueError("collective / TIM_AMD_DP_COLLECTIVE = %r: expected rs_ag, a2a or allreduce" % (want,))
            return want
        try:
            backend = str(dist.get_backend(self.pg))
        except Exception:  # noqa: BLE001
            backend = ""
        return "rs_ag" if (self.wire_dtype == torch.float32 and "nccl" in backend) else "a2a"

    def _preflight(self):
        """Everything about the preferred path that can fail on THIS rank without entering a collective: the A/B switch
        in the environment, the collectives' presence in this torch build, the wire dtype.  Raises or returns False for
        'not here'; tests replace it on one rank to force a rank-asymmetric refusal."""
        want = self._preferred()
        if want == "allreduce":
            return False
        if not hasattr(dist, "all_gather_into_tensor"):
            return False
        if want == "rs_ag":
            return hasattr(dist, "reduce_scatter_tensor") and self.wire_dtype == torch.float32
        if not hasattr(dist, "all_to_all_single"):
            return False
        return self.wire_dtype in (torch.float32, torch.bfloat16, torch.float16)

    def _agree(self, ok, dev):
        """logical AND of `ok` over the group (one small all-reduce; the only collective every backend runs)"""
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(flag.item() >= 0.5)

    @torch.no_grad()
    def _choose_collective(self, dev):
        """-> "a2a" | "allreduce", the same answer on every rank, for the life of the wrapper.

        A rank must never switch collectives on its own: peers that did not see its error are still inside all_to_all /
        all_gather, and a clean crash would become a collective mismatch (a hang, or wrongly summed buckets).  So (1) every
        rank runs its local preflight and the group ANDs the results - a refusal anywhere sends EVERYBODY to all_reduce
        before any all-to-all was issued; (2) the group runs one probe exchange on a small vector with a known answer
        through the very code path the step uses; an exception or a wrong mean on any rank is ANDed again.  After that
        `_exchange` has no fallback: an error in a step's exchange propagates (out of memory included) - the job stops
        instead of diverging."""
        W = self.world
        n = 64 * W + 24                      # not a multiple of 8 W: the staged (padded) form; 64 W alone = the zero-copy form
        want = "allreduce"
        self._wanted = None                  # what THIS rank asked for (None: its preference itself was invalid), kept for the warning
        try:
            want = self._wanted = self._preferred()
            ok = bool(self._preflight())
            if ok:
                # everything the probe allocates is allocated HERE, before the first collective: a rank-local failure (out of
                # memory included) is then ANDed across the group like any other refusal instead of leaving the peers inside
                # a collective this rank never enters
                rs = want == "rs_ag"
                self._staging(n, dev, rs=rs)
                self._staging(64 * W, dev, recv_only=self.wire_dtype == torch.float32, rs=rs)
        except Exception as e:  # noqa: BLE001  (a preflight that raises is a refusal, with the reason kept for the warning)
            ok, self._why = False, "preflight: %s" % str(e)[:200]
        # ranks that prefer DIFFERENT forms (an environment variable set on one of them) must not probe different collectives
        # against each other: the group agrees on one code (MIN over ranks: allreduce 0 < a2a 1 < rs_ag 2), everybody who
        # wanted something else counts as a refusal
        code = {"allreduce": 0.0, "a2a": 1.0, "rs_ag": 2.0}[want if ok else "allreduce"]
        flag = torch.tensor([code], dtype=torch.float32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
        low = int(round(flag.item()))
        if low == 0 or low != int(code):
            # somebody refused, or the ranks want different forms: one more agreement so that EVERY rank leaves here (the
            # ranks whose code equals the minimum cannot know whether all of them did)
            self._agree(False, dev)
            self._stage.clear()
            if not self._why and low != int(code):
                self._why = "the ranks prefer different exchange forms"
            return "allreduce"
        if not self._agree(True, dev):
            self._stage.clear()
            if not self._why:
                self._why = "the ranks prefer different exchange forms"
            return "allreduce"
        ok = True
        self.collective = want               # (the probe runs through the very code path the step uses)
        try:
            for numel in (n, 64 * W):
                # bounded values (at most 63 on every rank, whatever W): exact on a 16-bit wire too - an unbounded ramp
                # overflowed fp16 from W = 32 on and sent the group to all_reduce with a misleading "wrong mean"
                ramp = (torch.arange(numel, dtype=torch.float32, device=dev) % 64.0)
                probe = ramp * ((self.rank + 1) / float(W))
                self._exchange(probe)
                want_v = ramp * ((W + 1) / (2.0 * W))
                tol = 0.0 if self.wire_dtype == torch.float32 else 2.0 ** -7
                if not bool(((probe - want_v).abs() <= tol * want_v.abs() + 1e-4).all()):   # (fp32: a few ulps of 63 when W does not divide)
                    ok, self._why = False, "probe exchange returned a wrong mean"
        except Exception as e:  # noqa: BLE001  (any failure of this rank's probe is a refusal the group hears about)
            ok, self._why = False, "probe exchange: %s" % str(e)[:200]
        self._stage.clear()
        self.bytes_on_wire = 0
        return want if self._agree(ok, dev) else "allreduce"

    def _exchange(self, flat):
        """flat (fp32, 1-D) <- mean over ranks, by the form the group agreed on at construction (`self.collective`):
        reduce-scatter + all-gather, all-to-all + fp32 sum + all-gather on `wire_dtype`, or one fp32 all-reduce.  No per-rank
        fallback (see _choose_collective)."""
        W, n = self.world, flat.numel()
        if self._stub:
            # measurement mode (tools/dp_graph_check.py, TIM_AMD_DP_STUB=1): everything the wrapper does around an exchange - the
            # comm-stream fork and join, the events, the per-layer hooks - with the collectives replaced by ONE one-element
            # kernel: what is left of the wrapper's single-GPU cost when no byte is copied.  Never a training mode.
            flat[:1].add_(0.0)
            return
        if self.collective == "allreduce":
            dist.all_reduce(flat, group=self.pg)
            flat.mul_(1.0 / W)
            self.bytes_on_wire += 2 * 2 * (n // W) * (W - 1) * 4
            return
        if self.collective == "rs_ag":
            return self._exchange_rs(flat, W, n)
        return self._exchange_a2a(flat, W, n)

    def _exchange_rs(self, flat, W, n):
        """reduce-scatter (RCCL adds the W contributions of chunk r in fp32 on their way to rank r) + all-gather of the W
        means: SURVEY 8e's pattern as two library collectives - no receive buffer, no local sum pass, and (unlike the
        send / receive pairs behind all_to_all_single, which hang or crash hipStreamEndCapture on this stack:
        profiles/r05_rccl_capture_probe.txt) capturable in a HIP graph."""
        if self.wire_dtype == torch.float32 and n % (8 * W) == 0 and flat.is_contiguous():
            per = n // W
            st = self._staging(n, flat.device, recv_only=True, rs=True)
            dist.reduce_scatter_tensor(st["shard"], flat, group=self.pg)
            st["shard"].mul_(1.0 / W)
            dist.all_gather_into_tensor(flat, st["shard"], group=self.pg)
            self.bytes_on_wire += 2 * 2 * per * (W - 1) * 4
            return
        st = self._staging(n, flat.device, rs=True)
        per = st["per"]
        st["send"][:n].copy_(flat)
        dist.reduce_scatter_tensor(st["shard"], st["send"], group=self.pg)
        st["shard"].mul_(1.0 / W)
        dist.all_gather_into_tensor(st["send"], st["shard"], group=self.pg)
        flat.copy_(st["send"][:n])
        self.bytes_on_wire += 2 * 2 * per * (W - 1) * st["send"].element_size()   # (a 16-bit wire is summed in 16 bits here: use a2a for that)

    def _exchange_a2a(self, flat, W, n):
        if self.wire_dtype == torch.float32 and n % (8 * W) == 0 and flat.is_contiguous():
            # fp32 on the wire and a range that splits into W 32-byte-aligned chunks (bucket starts and sizes are multiples of
            # 64 elements, so this is every range when W divides 8): no staging copies - all-to-all out of the bucket,
            # all-gather into it; the only local memory pass is the fp32 sum
            per = n // W
            st = self._staging(n, flat.device, recv_only=True)
            dist.all_to_all_single(st["recv"], flat, group=self.pg)
            self._reduce_chunks(st, W, per, flat.is_cuda)
            dist.all_gather_into_tensor(flat, st["shard"], group=self.pg)
            self.bytes_on_wire += 2 * 2 * per * (W - 1) * 4
            return
        st = self._staging(n, flat.device)
        per = st["per"]
        st["send"][:n].copy_(flat)                                             # 1. narrow
        dist.all_to_all_single(st["recv"], st["send"], group=self.pg)          # 2. chunk r of every rank -> rank r
        self._reduce_chunks(st, W, per, flat.is_cuda)                          # 3. fp32 accumulation
        dist.all_gather_into_tensor(st["send"], st["shard"], group=self.pg)    # 4. (send is free again: reuse it)
        flat.copy_(st["send"][:n])                                             # 5. widen
        esz = st["send"].element_size()
        self.bytes_on_wire += 2 * 2 * per * (W - 1) * esz   # two phases, sent + received, W-1 peers of `per` elements

    def _comm_stream(self, dev):
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=dev)
        return self._comm

    def _on_bucket(self, name, flat, ready=None, members=None):
        if not self.active:
            return
        if not self.sync:
            self._accumulating = True     # this pass's gradients go into p.grad through autograd, un-exchanged
            return
        if self._accumulating and members:
            # earlier passes under no_sync() left local sums in p.grad: fold them into this bucket before the exchange and
            # clear p.grad, so that autograd ASSIGNS the averaged total (mean over ranks of g_1 + ... + g_k) instead of adding the
            # average of the last microbatch to a local sum
            ps = [(p, v) for p, v in members if p.grad is not None]
            if ps:
                if ready is not None and flat.is_cuda:
                    torch.cuda.current_stream().wait_event(ready)   # the side-stream weight gradients of this bucket
                with torch.no_grad():
                    torch._foreach_add_([v for _, v in ps], [p.grad for p, _ in ps])
                for p, _ in ps:
                    p.grad = None
        # consecutive buckets are contiguous in memory (same storage, ascending addresses): collect them into one range
        if self._pending and self._pending[-1][0].data_ptr() + self._pending[-1][0].numel() * 4 != flat.data_ptr():
            self._flush()      # (a bucket from elsewhere: exchange what is pending on its own)
        self._pending.append((flat, ready))
        if len(self._pending) >= self.buckets_per_exchange:
            self._flush()

    def _flush(self):
        if not self._pending:
            return
        first = self._pending[0][0]
        n = sum(f.numel() for f, _ in self._pending)
        flat = torch.as_strided(first, (n,), (1,))     # the range first .. last of the shared base buffer
        readies = [r for _, r in self._pending if r is not None]
        self._pending = []
        self._exchange_async(flat, readies)

    def _exchange_async(self, flat, readies=()):
        if flat.is_cuda:
            comm = self._comm_stream(flat.device)
            comm.wait_stream(torch.cuda.current_stream())
            for ready in readies:
                comm.wait_event(ready)  # weight gradients written on the side stream
            with torch.cuda.stream(comm):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(comm)
                self._exchange(flat)
                if self.timing:
                    e1.record(comm)
                    self.comm_events.append((e0, e1))
            flat.record_stream(comm)
        else:  # gloo / CPU tests of the bucket logic
            self._exchange(flat)

    def _on_encoder_done(self):
        if self.active and self.sync:
            self._flush()
            self._accumulating = False
        if self.active and self._comm is not None:
            torch.cuda.current_stream().wait_stream(self._comm)

    # ---- the few parameters outside the encoder Function -------------------------------------------
    def _on_small_grad(self, p):
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._reduce_small)

    def _reduce_small(self):
        """time MLP / DRLoc MLP gradients: gathered into one persistent flat buffer (one fused copy each way, no cat) and
        exchanged like a bucket; runs once, when the backward pass has finished"""
        self._callback_queued = False
        if not (self.active and self.sync):
            return
        ps = [p for p in self._small if p.grad is not None]
        if not ps:
            return
        n = sum(p.numel() for p in ps)
        dev = ps[0].grad.device
        if self._small_flat is None or self._small_flat.numel() != n or self._small_flat.device != dev:
            self._small_flat = torch.empty(n, dtype=torch.float32, device=dev)
        views, off = [], 0
        for p in ps:
            views.append(self._small_flat[off:off + p.numel()].view_as(p.grad))
            off += p.numel()
        grads = [p.grad for p in ps]
        if dev.type == "cuda":
            comm = self._comm_stream(dev)
            comm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(comm):
                torch._foreach_copy_(views, grads)
                self._exchange(self._small_flat)
                torch._foreach_copy_(grads, views)
            torch.cuda.current_stream().wait_stream(comm)
        else:
            torch._foreach_copy_(views, grads)
            self._exchange(self._small_flat)
            torch._foreach_copy_(grads, views)

    # ---- measurement hooks (bench.py --gpus N) --------------------------------------------------------
    def begin_step_timing(self):
        self.timing = True
        self.comm_events = []
        self.bytes_on_wire = 0

    def end_step_timing(self):
        """-> (milliseconds the comm stream spent in this step's bucket exchanges, bytes sent + received per rank)"""
        self.timing = False
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.comm_events)
        return ms, self.bytes_on_wire

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def allreduce_buckets_reference(bucket_tensors, world, group=None):
    """The exact fp32 mean (what the exchange approximates on a bf16 wire): in place; used by the tests."""
    for t in bucket_tensors:
        dist.all_reduce(t, group=group)
        t.div_(world)
