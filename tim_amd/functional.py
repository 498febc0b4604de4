"""Host-side sequencing of the TIM hot path over libtimhip's C ABI.

Two autograd Functions mirror the two entry points the reference loops call
(rec train.py:198,209-215): `time_mlp` (tim.py:66-74,181-182) and `encoder`
(tim.py:147-172).  All arithmetic runs in the HIP library; torch supplies device
buffers, the current stream and autograd bookkeeping only.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib as L
from ._lib import call, ptr


def _ru(x, m=64):
    return (x + m - 1) // m * m


def _stream():
    return torch.cuda.current_stream().cuda_stream


# graph-safe dropout (include/timhip.h: timhip_dropout_salt): one 64-bit word in device memory that every dropout kernel adds
# to its launch-time seed.  Process-wide, like the library's pointer to it (one process per GPU).
_SALT = {"word": None, "epoch": 0}
_SALT_STEP = 0x51B54A32D192ED03  # < 2^63: int64 adds wrap on the device, which is all a salt needs


def graph_safe_dropout(dev, enable=True):
    """Register (or drop) the device-side dropout salt.  While registered, `Runtime.next_seed` advances the salt on the
    device instead of changing the seed it hands to the launches - the form a captured HIP graph needs.  One encoder
    forward must be followed by its own backward before the next training forward (the usual step structure): the
    backward regenerates its masks from the salt the forward left behind."""
    if not enable:
        call("timhip_dropout_salt", None)
        _SALT["word"] = None
        return None
    w = _SALT["word"]
    if w is not None and w.device != torch.device(dev):
        raise RuntimeError("the dropout salt lives on %s; one process drives one GPU" % w.device)
    if w is None:
        w = torch.zeros(1, dtype=torch.int64, device=dev)
        call("timhip_dropout_salt", ptr(w))
        _SALT["word"] = w
    return w


class Runtime:
    """Per-model state that is not a parameter: precision, operand-dtype working copies of
    the weights (plain and transposed, refreshed when a parameter's version changes),
    the Philox step counter, and the gradient-bucket hook used by data parallelism."""

    def __init__(self, precision="fp16"):
        if precision not in L.PRECISIONS:
            raise ValueError("precision must be one of %s" % list(L.PRECISIONS))
        self.precision_name = precision
        self.prec = L.PRECISIONS[precision]
        # operand storage: bf16 / fp16 for the 16-bit MFMA modes, fp32 otherwise (bf16x3 splits on the fly)
        self.op_dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(precision, torch.float32)
        self.h16 = self.prec in L.H16
        # fp16: 11-bit operands bring the 6-layer logits within 1e-3 of the fp32 reference only if the two small sites that
        # dominate the error budget - the time MLP (its LayerNorm amplifies) and the classification heads (they write the
        # logits) - keep more bits (oracle site analysis).  Those two run with SPLIT operands: every
        # fp32 value as hi + lo fp16 column blocks, one fp16 GEMM over the tripled contraction length (timhip_split3_many).
        self.split = precision == "fp16"
        # Opt-in margin modes (round 3), TIM_AMD_SPLIT_LAYER_WEIGHTS = none (default) | out | all, or `rt.layer_split = (...)`
        # before the first forward: the WEIGHTS of the encoder layers' forward GEMMs as hi + lo halves - a weight's rounding error
        # is the same for every token and survives the attention average (an activation's does not).  The product runs
        # [x | x] [w_hi | w_lo]^T over K = 2K with the activation operand read twice (TimEpi.a_wrap_k).  Measured on C2a, B = 64
        # (DESIGN.md section 6): none 8.1e-4 max logit error; "out" (the out-projection: largest single term, smallest GEMM)
        # 7.2e-4 at +2 % of the step; "all" 5.3e-4 at +15 %.
        sel = os.environ.get("TIM_AMD_SPLIT_LAYER_WEIGHTS", "none") if self.split else "none"
        modes = {"none": (), "out": ("out",), "all": ("in", "out", "l1", "l2")}
        if sel not in modes:
            raise ValueError("TIM_AMD_SPLIT_LAYER_WEIGHTS=%r: expected one of %s" % (sel, sorted(modes)))
        self._layer_split = ()
        self._split_warned = False
        self.layer_split = modes[sel]
        self._wsplit = {}
        self._wsparams = {}  # id(param) -> weakref: every weight this runtime has split
        # fp16 backward: gradient operands are stored times a power of two chosen per backward pass from the incoming
        # cotangents (timhip_grad_scale): S * max|cotangent| ~ grad_scale_target.  16 leaves a factor 4096 of headroom below
        # the fp16 maximum for gradients that grow along the backward chain (LayerNorm's 1/std) and 2^-18 of the largest
        # cotangent before values go subnormal
        self.grad_scale_target = 16.0
        # fp16: the residual part of the backward's gradient stream between LayerNorms stays fp32 - as in the reference's AMP recipe,
        # whose residual gradient is fp32 (scripts/train.py:82,355-363).  TIM_AMD_GRAD_STREAM=16 (opt-in, round 4's default) carries
        # it 16-bit under the gradient scale: 40 MB less per LayerNorm-backward launch at C2a = -0.8 % of the step, at the price of
        # parameter gradients within 1.4e-3 instead of 8.7e-4 of the oracle's - more rounding than the reference applies there
        # (round-4 review: the wrong way to spend margin)
        gsel = os.environ.get("TIM_AMD_GRAD_STREAM", "fp32")
        if gsel not in ("16", "fp32"):
            raise ValueError("TIM_AMD_GRAD_STREAM=%r: expected 16 or fp32" % gsel)
        self.grad_stream16 = gsel == "16"
        self._gs_blocks = []   # this runtime's timhip_grad_scale blocks since the last grads_finite() (word 4 = non-finite flag)
        self._gs_captured = []  # ... and the blocks of captured (HIP-graph) backward passes nobody owns yet (strong references)
        self._gs_adopted = []   # ... weak references to the blocks a GraphedStep owns (adopt_captured)
        self._nf_acc = None
        self._wcache = {}
        self._wparams = {}  # id(param) -> weakref: every weight this runtime has cast
        # dropout stream: seeded from torch's generator (torch.manual_seed / args.seed select the run's masks, as they do in the
        # reference) and from the rank (data-parallel replicas draw different masks); TIM.dropout_rng_state() /
        # set_dropout_rng_state() let a checkpoint resume the sequence instead of replaying it
        try:
            import torch.distributed as _dist
            rank = _dist.get_rank() if _dist.is_available() and _dist.is_initialized() else 0
        except Exception:  # noqa: BLE001
            rank = 0
        self.seed = ((torch.initial_seed() & 0xFFFFFFFFFFFF) * 0x9E3779B1 + rank * 0x85EBCA6B + 0x5EED) & 0x7FFFFFFFFFFFFFFF
        self.step = 0
        self.last_seed = 0   # the Philox key of the most recent training forward (what timhip_dropout_mask reproduces)
        self.bucket_hook = None  # callable(bucket_name, flat_grad_tensor) -> None
        self.finish_hook = None  # callable() -> None, called at the end of the encoder backward
        # TIM_AMD_OVERLAP_WGRAD=1: the weight-gradient launch of layer l runs on a side stream, overlapping the data chain of
        # layer l-1.  Off by default since the layer's weight gradients became ONE grid that fills every block slot for its whole
        # duration: run concurrently it starves the data chain (the critical path) and the step is 2 % slower than in sequence
        # (measured); with four short launches per layer the overlap used to gain 3 %.
        self.overlap_wgrad = os.environ.get("TIM_AMD_OVERLAP_WGRAD", "0") == "1"
        self.separate_wgrad = os.environ.get("TIM_AMD_WGRAD_SEPARATE", "0") == "1"  # A/B: per-Linear weight-gradient launches
        self._aux = {}
        self.recast_every_forward = False

    def aux_stream(self, dev):
        st = self._aux.get(dev)
        if st is None:
            st = torch.cuda.Stream(device=dev)
            self._aux[dev] = st
        return st

    # ---- buffers -------------------------------------------------------------------------------
    def zeros_op(self, rows, cols, dev):
        return torch.zeros((rows, _ru(cols)), dtype=self.op_dtype, device=dev)

    def out_op(self, rows, cols, dev):
        """operand buffer [rows, ru(cols)] that a kernel is about to fill completely in its first `cols` columns: only a
        padded buffer (cols not a multiple of 64) needs the zero fill, for its padding columns"""
        if cols % 64 == 0:
            return torch.empty((rows, cols), dtype=self.op_dtype, device=dev)
        return torch.zeros((rows, _ru(cols)), dtype=self.op_dtype, device=dev)

    def empty_op(self, rows, cols, dev):
        assert cols % 64 == 0
        return torch.empty((rows, cols), dtype=self.op_dtype, device=dev)

    # ---- weight working copies -----------------------------------------------------------------
    def weight(self, p, transposed=False):
        """operand-dtype copy of a [N,K] fp32 weight: [N, ru(K)] or (transposed) [K, ru(N)].  Both copies are
        produced together (one pass over the fp32 master) whenever the parameter's version changed; when one
        weight is found stale, every stale weight this runtime has seen is refreshed in the same launch
        (after an optimizer step that is all of them: one kernel instead of one per weight)."""
        ent = self._wcache.get(id(p))
        # (same storage address and version = same contents on the same device: no separate device comparison on the fast path)
        if ent is None or ent[0] != (p.data_ptr(), p._version):
            self._wparams[id(p)] = weakref.ref(p)
            self._refresh(p.device)
            ent = self._wcache[id(p)]
        return ent[2] if transposed else ent[1]

    def _refresh(self, dev):
        items, keep, fresh = [], [], {}
        for key, ref in list(self._wparams.items()):
            q = ref()
            if q is None:
                self._wparams.pop(key, None)
                self._wcache.pop(key, None)
                continue
            if q.device != dev:
                continue
            ent = self._wcache.get(key)
            ver = (q.data_ptr(), q._version)
            if ent is not None and ent[0] == ver and ent[1].device == dev:
                continue
            N, K = q.shape
            src = q.detach()
            if src.dtype != torch.float32 or not src.is_contiguous():
                src = src.float().contiguous()
            if ent is not None and ent[1].device == dev and ent[1].shape == (N, _ru(K)):
                plain, tr = ent[1], ent[2]
            else:
                plain = torch.empty((N, _ru(K)), dtype=self.op_dtype, device=dev)
                tr = torch.empty((K, _ru(N)), dtype=self.op_dtype, device=dev)
            items.append(L.TimCastItem(ptr(src), ptr(plain), ptr(tr), N, K, plain.shape[1], tr.shape[1]))
            keep.append(src)
            fresh[key] = (ver, plain, tr)
        if items:
            arr = (L.TimCastItem * len(items))(*items)
            call("timhip_cast_weights", self.prec, C.cast(arr, C.c_void_p), len(items), _stream())
            self._wcache.update(fresh)

    # ---- opt-in weight-split mode of the encoder layers' forward GEMMs -----------------------------------------
    @property
    def layer_split(self):
        return self._layer_split

    @layer_split.setter
    def layer_split(self, keys):
        keys = tuple(keys)
        bad = [k for k in keys if k not in ("in", "out", "l1", "l2")]
        if bad:
            raise ValueError("layer_split: unknown Linear %r (expected a subset of in / out / l1 / l2)" % (bad,))
        if keys and not self.split:
            raise ValueError("layer_split is a margin mode of precision='fp16' only")
        self._layer_split = keys

    @property
    def split_outproj(self):          # (derived: never stale when layer_split is assigned after construction)
        return "out" in self._layer_split

    def layer_split_for(self, E, FF):
        """the split set this model can run: the wrapped-operand product (TimEpi.a_wrap_k) needs contraction lengths that are
        multiples of 64, i.e. E % 64 == 0 and FF % 64 == 0 (the model itself only asks for d_model % 32 == 0); otherwise plain
        weights, with one warning - not a TIMHIP_EUNSUPPORTED at the first layer"""
        if self._layer_split and (E % 64 or FF % 64):
            if not self._split_warned:
                import warnings
                warnings.warn("tim_amd: layer_split %r needs E %% 64 == 0 and FF %% 64 == 0 (E = %d, FF = %d): running with plain "
                              "16-bit layer weights" % (self._layer_split, E, FF))
                self._split_warned = True
            return ()
        return self._layer_split

    def layer_split_flags(self, E, FF):
        f = {"in": L.DESC_INPROJ_SPLIT, "out": L.DESC_OUTPROJ_SPLIT, "l1": L.DESC_L1_SPLIT, "l2": L.DESC_L2_SPLIT}
        return sum(f[k] for k in self.layer_split_for(E, FF))

    def weight_split(self, p, mode=1):
        """split copy of an fp32 weight [N, K] as [N, 3 ru(K)] 16-bit column blocks: mode 1 = [hi | hi | lo] (the weight side
        of a three-term split product: time MLP, heads), mode 0 = [hi | lo | hi] (its first two blocks are the weight side of
        the TWO-term product [x | x] [w_hi | w_lo]^T of the encoder layers' out-projection, TIMHIP_DESC_OUTPROJ_SPLIT).  As in
        `weight`, one stale copy refreshes every stale split copy of that mode in the same grouped launch."""
        key = (id(p), mode)
        ent = self._wsplit.get(key)
        ver = (p.data_ptr(), p._version)
        if ent is None or ent[0] != ver or ent[1].device != p.device:
            self._wsparams[key] = weakref.ref(p)
            self._refresh_split(p.device, mode)
            ent = self._wsplit[key]
        return ent[1]

    def _refresh_split(self, dev, mode):
        items, fresh = [], {}
        for key, ref in list(self._wsparams.items()):
            if key[1] != mode:
                continue
            q = ref()
            if q is None:
                self._wsparams.pop(key, None)
                self._wsplit.pop(key, None)
                continue
            if q.device != dev:
                continue
            ent = self._wsplit.get(key)
            ver = (q.data_ptr(), q._version)
            if ent is not None and ent[0] == ver and ent[1].device == dev:
                continue
            N, K = q.shape
            src = _f32c(q)
            buf = ent[1] if ent is not None and ent[1].device == dev and ent[1].shape == (N, 3 * _ru(K)) else \
                torch.empty((N, 3 * _ru(K)), dtype=self.op_dtype, device=dev)
            items.append((src, N, K, K, buf))
            fresh[key] = (ver, buf, src)
        if items:
            self.split3(items, mode=mode)
            self._wsplit.update(fresh)

    def split3(self, items, mode, relu=False):
        """items: [(src fp32 [rows, cols] with row stride lds, rows, cols, lds, dst [rows, 3 ru(cols)])]"""
        items = [it for it in items if it[1] > 0]
        for i0 in range(0, len(items), 6):
            grp = items[i0:i0 + 6]
            call("timhip_split3_many", self.prec, len(grp), _parr([g[0] for g in grp]), _iarr([g[1] for g in grp]),
                 _iarr([g[2] for g in grp]), _iarr([g[3] for g in grp]), _parr([g[4] for g in grp]),
                 _iarr([g[4].shape[1] for g in grp]), mode, 1 if relu else 0, _stream())

    def invalidate_weights(self):
        """Force the operand copies to be rebuilt.  The copies are keyed on (data_ptr, version counter) of the parameter:
        optimizer steps and every in-place op on the parameter bump the version and are picked up automatically; writes
        through `p.data` (custom init, EMA / weight surgery code, `load_state_dict` goes through copy_ and IS tracked) do not
        bump it - call this (also exported as `TIM.invalidate_weights()`) after such a write, or set
        `rt.recast_every_forward = True` to rebuild the copies at every training forward (one grouped launch, ~0.1 ms at
        C2a)."""
        for k, ent in list(self._wcache.items()):
            self._wcache[k] = (None, ent[1], ent[2])
        for k, ent in list(self._wsplit.items()):
            self._wsplit[k] = (None, ent[1], ent[2])

    def next_seed(self):
        self.step += 1
        self.last_seed = self._next_seed()
        return self.last_seed

    def _next_seed(self):
        if _SALT["word"] is not None:
            # graph-safe mode: the launch-time seed stays fixed, the per-step part lives in device memory and is advanced
            # by a (capturable) device-side add, so a replayed graph draws fresh masks every time
            _SALT["word"].add_(_SALT_STEP)
            _SALT["epoch"] += 1
            return (self.seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        return (self.seed * 0x9E3779B97F4A7C15 + self.step * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def grad_scale(self, cotangents, dev, out=None):
        """fp16 only: device tensor {S, 1/S, 0, 0} for one backward pass (None in the other modes), S chosen on the device
        from the largest |cotangent| - no host synchronisation."""
        if self.prec != L.PREC_F16:
            return None
        dev = torch.device(dev)
        # {S, 1/S, scratch, scratch, non-finite flag, 0, 0, 0}; `out`: a block the caller has already zero-filled
        gs = out if out is not None else torch.zeros(8, dtype=torch.float32, device=dev)
        if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # a block of a captured step lives as long as the graph and is rewritten by every replay: watched for good
            self._gs_captured.append(gs)
        else:
            # many backward passes without a reader: fold the OLDER blocks on the device (no sync) BEFORE this pass's block joins
            # the list - its flag word is still zero here (its kernels are issued below), folding it would lose the pass's flag
            if len(self._gs_blocks) >= 16:
                self._fold_flags(captured=False)
            self._gs_blocks.append(gs)
        cots = [c for c in cotangents if c is not None and c.numel() > 0]
        if not cots:
            gs[:2] = 1.0
            return gs
        for i0 in range(0, len(cots), 8):   # (more than 8 cotangent tensors: the last group decides - never the case for TIM)
            grp = cots[i0:i0 + 8]
            call("timhip_grad_scale", _parr(grp), (C.c_longlong * len(grp))(*[c.numel() for c in grp]), len(grp),
                 float(self.grad_scale_target), ptr(gs), _stream())
        return gs

    def _fold_flags(self, captured=True):
        if self._gs_blocks:
            f = torch.stack([g[4] for g in self._gs_blocks]).view(torch.int32).ne(0).any()
            self._nf_acc = f if self._nf_acc is None else (self._nf_acc | f)
            self._gs_blocks = []
        if captured:   # (not consumed: the next replay zeroes and rewrites them)
            self._gs_adopted = [r for r in self._gs_adopted if r() is not None]   # blocks of dropped graphs: no longer watched
            live = list(self._gs_captured) + [r() for r in self._gs_adopted]
            live = [g for g in live if g is not None]
            if live:
                f = torch.stack([g[4] for g in live]).view(torch.int32).ne(0).any()
                self._nf_acc = f if self._nf_acc is None else (self._nf_acc | f)

    def adopt_captured(self, since=0):
        """Hand the gradient-scale blocks of the backward passes captured so far (from list position `since`) to the caller (`GraphedStep` after its
        capture): the runtime keeps WEAK references from here on, so the blocks - and the watch on their non-finite flags -
        end with the object that owns the graph.  Blocks nobody adopts (a bare `torch.cuda.graph` capture) stay watched until
        `forget_captured()`."""
        blocks, self._gs_captured = self._gs_captured[since:], self._gs_captured[:since]
        self._gs_adopted += [weakref.ref(b) for b in blocks]
        return blocks

    def forget_captured(self):
        """Stop watching the blocks of every captured backward pass (a graph was dropped or is about to be re-captured: its
        last replay may have left a flag set that nothing rewrites any more - watched for good it would make every later
        grads_finite() False and a `if rt.grads_finite(): opt.step()` loop skip every step)."""
        self._gs_captured = []
        self._gs_adopted = []

    def grads_finite(self, reset=True):
        """False iff a weight / bias gradient written since the last call (by this runtime's backward passes) was inf or nan -
        the fp16 mode's counterpart of GradScaler's inf check (reference scripts/train.py:351,357-363: skip the optimizer
        step).  The kernels that write the gradients OR one device word (include/timhip.h: timhip_grad_scale); nothing is
        synchronised until this call reads it.  Always True in the fp32 / bf16 modes (8 exponent bits: no overflow to watch).
        Under HIP-graph replay the words belong to the captured backward passes: every replay zeroes and rewrites them, so
        they describe the LATEST replay (reading does not clear them; eager passes in between are folded in as usual)."""
        self._fold_flags()
        if self._nf_acc is None:
            return True
        bad = bool(self._nf_acc.item())
        if reset:
            self._nf_acc = None
        return not bad

    # ---- thin op wrappers ------------------------------------------------------------------------
    def gemm(self, epi, A, B, M, N, K, out0, ld0, out1=None, ld1=0, bias=None, res=None, ldres=0, aux=None,
             ldaux=0, p_drop=0.0, seed=0, site=0, splitk=1, mask=None, ldmask=0, ln=None, acc_scale=None, rep=0, a_wrap_k=0):
        """ln = (stats[M,2], gamma[N], beta[N]): EPI_DROP_RES_F32 takes LayerNorm(res) as its residual (TimEpi.ln_*);
        acc_scale: device pointer (int) of a scalar multiplied into the accumulators, or None"""
        if M == 0 or N == 0:
            return
        st_, g_, b_ = ln if ln is not None else (None, None, None)
        e = L.TimEpi(ptr(out0), ptr(out1), ptr(bias), ptr(res), ptr(aux), ld0, ld1, ldres, ldaux,
                     float(p_drop), site, seed, ptr(mask), ldmask, rep, ptr(st_), ptr(g_), ptr(b_), acc_scale, a_wrap_k, 0)
        call("timhip_gemm_nt", self.prec, epi, ptr(A), A.stride(0), ptr(B), B.stride(0), M, N, K,
             C.byref(e), splitk, _stream())

    def wgrad(self, dY, Nout, X, Kout, M, dW, db, out_scale=None):
        """dW[Nout,Kout] += dY[:M,:Nout]^T X[:M,:Kout]; db += colsum(dY)   (out_scale: device pointer of a factor on both)"""
        if M == 0:
            return
        nbytes = L.load().timhip_wgrad_workspace_bytes(self.prec, Nout, Kout, M)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dY.device)
        call("timhip_wgrad", self.prec, ptr(dY), dY.stride(0), Nout, ptr(X), X.stride(0), Kout, M, ptr(dW),
             ptr(db), ptr(ws), nbytes, out_scale, _stream())

    def gemm_many(self, epi, items, acc_scale=None):
        """items: [dict(A, B, M, N, K, out0, ld0, bias=None, res=None, ldres=0)] - independent small problems with the same
        epilogue; with 16-bit operands up to six go out as one grouped launch (timhip_gemm_nt_group), otherwise one launch each"""
        items = [it for it in items if it["M"] > 0 and it["N"] > 0]
        if not self.h16 or len(items) < 2 or os.environ.get("TIM_AMD_NO_GEMM_GROUP", "0") == "1":  # (A/B switch)
            for it in items:
                self.gemm(epi, it["A"], it["B"], it["M"], it["N"], it["K"], it["out0"], it["ld0"], bias=it.get("bias"),
                          res=it.get("res"), ldres=it.get("ldres", 0), acc_scale=acc_scale, rep=it.get("rep", 0))
            return
        for i0 in range(0, len(items), 6):
            grp = items[i0:i0 + 6]
            arr = (L.TimGemmItem * len(grp))()
            for a, it in zip(arr, grp):
                a.A, a.B = ptr(it["A"]), ptr(it["B"])
                a.lda, a.ldb = it["A"].stride(0), it["B"].stride(0)
                a.M, a.N, a.K, a.reserved = it["M"], it["N"], it["K"], it.get("rep", 0)
                a.e = L.TimEpi(ptr(it["out0"]), None, ptr(it.get("bias")), ptr(it.get("res")), None, it["ld0"], 0,
                               it.get("ldres", 0), 0, 0.0, 0, 0, None, 0, 0, None, None, None, acc_scale)
            call("timhip_gemm_nt_group", self.prec, epi, C.cast(arr, C.c_void_p), len(grp), _stream())

    def wgrad_many(self, items, out_scale=None):
        """items: [(dY, Nout, X, Kout, M, dW, db), ...] - accumulate every weight gradient; with 16-bit operands the items that
        share M go out as one grouped launch (front end, heads: many small GEMMs that each would need their own split-K + reduce)"""
        if not self.h16:
            for dY, Nout, X, Kout, M, dW, db in items:
                self.wgrad(dY, Nout, X, Kout, M, dW, db, out_scale)
            return
        by_m = {}
        for it in items:
            dY, Nout, X, Kout, M, dW, db = it
            if M == 0:
                continue
            if (Nout * Kout) % 4:
                self.wgrad(dY, Nout, X, Kout, M, dW, db, out_scale)
            else:
                by_m.setdefault(M, []).append((dY, Nout, X, Kout, dW, db))
        for M, grp in by_m.items():
            for i in range(0, len(grp), 8):
                self.wgrad_group(grp[i:i + 8], M, accumulate=True, out_scale=out_scale)

    def wgrad_group(self, items, M, accumulate=True, out_scale=None):
        """items: [(dY, Nout, X, Kout, dW, db|None), ...] sharing M - the weight gradients of several Linear layers as one
        launch (bf16; timhip_wgrad_group)"""
        arr = (L.TimWgradItem * len(items))(*[L.TimWgradItem(ptr(dY), ptr(X), ptr(dW), ptr(db), dY.stride(0), X.stride(0),
                                                             Nout, Kout) for dY, Nout, X, Kout, dW, db in items])
        pa = C.cast(arr, C.c_void_p)
        nbytes = L.load().timhip_wgrad_group_workspace_bytes(self.prec, pa, len(items), M)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=items[0][0].device)
        call("timhip_wgrad_group", self.prec, pa, len(items), M, 1 if accumulate else 0, ptr(ws), nbytes, out_scale, _stream())

    def ln_fwd(self, y, rows, cols, act, w, b, xf=None, ldx=0, xt=None, ldt=0, stats=None):
        call("timhip_layernorm_fwd", self.prec, ptr(y), rows, cols, y.stride(0), act, ptr(w), ptr(b), ptr(xf), ldx,
             ptr(xt), ldt, ptr(stats), _stream())

    def ln_bwd(self, dx, y, stats, rows, cols, act, w, dyf=None, dyt=None, dgamma=None, dbeta=None, t_scale=None):
        call("timhip_layernorm_bwd", self.prec, ptr(dx), dx.stride(0), ptr(y), y.stride(0), ptr(stats), rows, cols,
             act, ptr(w), ptr(dyf), 0 if dyf is None else dyf.stride(0), ptr(dyt),
             0 if dyt is None else dyt.stride(0), 0.0, 0, 0, ptr(dgamma), ptr(dbeta), t_scale, _stream())


def _iarr(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def _parr(tensors):
    return (C.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


def _f32c(t):
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _require_gpu(t, what):
    if not t.is_cuda:
        raise L.TimHipError("%s: the TIM hot path runs on the MI355X HIP kernels only; got a %s tensor "
                            "(there is no CPU fallback)" % (what, t.device))
    L.load()


# ==================================================================================================
# time MLP   (tim.py:66-74)
# ==================================================================================================
class TimeMlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rt, times, w0, b0, w2, b2, w4, b4, lnw, lnb):
        _require_gpu(times, "time_mlp")
        dev = times.device
        d = w0.shape[0]
        t2 = _f32c(times).reshape(-1, 2)
        R = t2.shape[0]
        ldd = _ru(d)
        w0c, b0c, b2c, b4c, lnwc, lnbc = [_f32c(t) for t in (w0, b0, b2, b4, lnw, lnb)]
        u3 = torch.empty((R, d), dtype=torch.float32, device=dev)
        if rt.split:
            # fp16 model: split operands (Runtime.__init__).  h1 / h2 are kept as [hi | lo | hi] blocks; their first block is
            # the plain fp16 operand the backward reads (row stride 3 ldd)
            # (layer 1 and the epilogue of layer 2 write the [hi | lo | hi] blocks themselves: no fp32 round trip, no split launches)
            h1 = torch.empty((R, 3 * ldd), dtype=rt.op_dtype, device=dev)
            call("timhip_time_l1_fwd_split3", rt.prec, ptr(t2), R, d, ptr(w0c), ptr(b0c), ptr(h1), ldd, _stream())
            h2 = (torch.empty if d == ldd else torch.zeros)((R, 3 * ldd), dtype=rt.op_dtype, device=dev)   # (padding columns stay zero)
            rt.gemm(L.EPI_RELU_SPLIT3_T, h1, rt.weight_split(w2), R, d, 3 * ldd, h2, 3 * ldd, ld1=ldd, bias=b2c, rep=3)
            rt.gemm(L.EPI_STORE_F32, h2, rt.weight_split(w4), R, d, 3 * ldd, u3, d, bias=b4c, rep=3)
        else:
            h1 = rt.out_op(R, d, dev)
            call("timhip_time_l1_fwd", rt.prec, ptr(t2), R, d, ptr(w0c), ptr(b0c), ptr(h1), ldd, _stream())
            h2 = rt.out_op(R, d, dev)
            rt.gemm(L.EPI_RELU_T, h1, rt.weight(w2), R, d, d, h2, ldd, bias=b2c)
            rt.gemm(L.EPI_STORE_F32, h2, rt.weight(w4), R, d, d, u3, d, bias=b4c)
        te = torch.empty((R, d), dtype=torch.float32, device=dev)
        stats = torch.empty((R, 2), dtype=torch.float32, device=dev)
        rt.ln_fwd(u3, R, d, 1, lnwc, lnbc, xf=te, ldx=d, stats=stats)
        ctx.rt = rt
        ctx.shape = tuple(times.shape)
        ctx.save_for_backward(t2, h1, h2, u3, stats, w0, w2, w4, lnw)
        return te.view(*times.shape[:-1], d)

    @staticmethod
    def backward(ctx, d_te):
        rt = ctx.rt
        t2, h1, h2, u3, stats, w0, w2, w4, lnw = ctx.saved_tensors
        dev = t2.device
        R, d = u3.shape
        ldd = _ru(d)
        g = _f32c(d_te).reshape(R, d)
        # the eight (accumulated-into) gradient tensors as views of ONE zero-filled buffer; its last 8 words are the
        # gradient-scale block of the fp16 mode (one fill launch for both)
        shapes = [(d, 2), (d,), (d, d), (d,), (d, d), (d,), (d,), (d,)]
        sizes = [(int(torch.Size(sh).numel()) + 3) // 4 * 4 for sh in shapes]
        flat = torch.zeros(sum(sizes) + 8, dtype=torch.float32, device=dev)
        gs = rt.grad_scale([g], dev, out=flat[sum(sizes):])
        gs_in, gs_out = (ptr(gs), ptr(gs) + 4) if gs is not None else (None, None)
        views, off = [], 0
        for sh, n in zip(shapes, sizes):
            views.append(flat[off:off + int(torch.Size(sh).numel())].view(sh))
            off += n
        dw0, db0, dw2, db2, dw4, db4, dlnw, dlnb = views
        du3 = rt.out_op(R, d, dev)
        rt.ln_bwd(g, u3, stats, R, d, 1, _f32c(lnw), dyt=du3, dgamma=dlnw, dbeta=dlnb, t_scale=gs_in)
        du2 = rt.out_op(R, d, dev)
        rt.gemm(L.EPI_DRELU_T, du3, rt.weight(w4, True), R, d, d, du2, ldd, aux=h2, ldaux=h2.stride(0))
        rt.wgrad_many([(du3, d, h2, d, R, dw4, db4), (du2, d, h1, d, R, dw2, db2)], out_scale=gs_out)
        du1 = rt.out_op(R, d, dev)
        rt.gemm(L.EPI_DRELU_T, du2, rt.weight(w2, True), R, d, d, du1, ldd, aux=h1, ldaux=h1.stride(0))
        d_times = torch.empty((R, 2), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        call("timhip_time_l1_bwd", rt.prec, ptr(t2), R, d, ptr(_f32c(w0)), ptr(du1), ldd, ptr(dw0), ptr(db0),
             ptr(d_times), gs_out, _stream())
        if d_times is not None:
            d_times = d_times.view(ctx.shape)
        return None, d_times, dw0, db0, dw2, db2, dw4, db4, dlnw, dlnb


# ==================================================================================================
# encoder: feature encoding + L layers + heads   (tim.py:147-172)
# ==================================================================================================
class EncoderPlan:
    """Token-row table and head slices for one (config, T, Nv, Na).  Mirrors the concatenation
    order of encodings.py:190-250 and the tail slicing of head.py:17-38 (det head.py:27-46)."""

    def __init__(self, cfg, T, nv, na):
        nf, F = cfg.num_feats, cfg.F
        det = cfg.variant == "detection"
        rows = []  # (kind, src, te_row, mod)
        self.cls_names = []
        self.mod_names = []
        self.embedders = []  # (name, e-slot)

        def cls_idx(name):
            if name not in self.cls_names:
                self.cls_names.append(name)
            return self.cls_names.index(name)

        av = cfg.input_modality == "audio_visual"
        if av:
            self.mod_names = ["visual_modality_encoding", "audio_modality_encoding"]
            self.embedders = [("visual", 0), ("audio", 1)]
            rows += [(0, s, s, 0) for s in range(nf)]
            rows += [(2, s, nf + s, 1) for s in range(nf)]
            nq_te = T - 2 * nf
            if "visual" in cfg.data_modality and nv > 0:
                groups = (["visual_verb_cls", "visual_noun_cls"] if (cfg.include_verb_noun and not det) else []) \
                    + ["visual_action_cls"]
                for g in groups:
                    rows += [(1, cls_idx(g), 2 * nf + j, 0) for j in range(nv)]
            if "audio" in cfg.data_modality and na > 0:
                rows += [(1, cls_idx("audio_action_cls"), 2 * nf + nq_te - na + j, 1) for j in range(na)]
        elif cfg.input_modality == "visual":
            self.embedders = [("visual", 0)]
            rows += [(0, s, s, -1) for s in range(nf)]
            nq = T - nf
            if det:
                groups = ["visual_action_cls"]
            else:
                groups = (["verb_cls", "noun_cls"] if cfg.include_verb_noun else []) + ["action_cls"]
            for g in groups:
                rows += [(1, cls_idx(g), nf + j, -1) for j in range(nq)]
        else:
            self.embedders = [("audio", 0)]
            rows += [(0, s, s, -1) for s in range(nf)]
            nq = T - nf
            g = "audio_action_cls" if det else "action_cls"
            rows += [(1, cls_idx(g), nf + j, -1) for j in range(nq)]
        self.rows = rows
        self.S = len(rows)
        self.F = F
        self.T = T
        S = self.S
        # heads: (output slot, parameter prefix, s0, n)
        heads = []
        nc = cfg.num_class
        if cfg.data_modality == "audio_visual":
            aud_start = S - na if na > 0 else S
            act_start = aud_start - nv
            vn = isinstance(nc, list) if det else isinstance(nc[0], list)
            if vn:
                if det:
                    heads += [("verb", "fc_visual_verb", act_start, nv), ("noun", "fc_visual_noun", act_start, nv)]
                else:
                    heads += [("verb", "fc_visual_verb", act_start - 2 * nv, nv),
                              ("noun", "fc_visual_noun", act_start - nv, nv)]
            heads += [("action", "fc_visual_action", act_start, nv), ("audio", "fc_audio_action", aud_start, S - aud_start)]
            self.reg = [("reg_visual", "fc_visual_action", act_start, nv),
                        ("reg_audio", "fc_audio_action", aud_start, S - aud_start)] if det else []
        elif cfg.data_modality == "visual":
            act_start = S - nv
            if isinstance(nc[0], list):
                if det:
                    heads += [("verb", "fc_visual_verb", act_start, nv), ("noun", "fc_visual_noun", act_start, nv)]
                else:
                    heads += [("verb", "fc_visual_verb", act_start - 2 * nv, nv),
                              ("noun", "fc_visual_noun", act_start - nv, nv)]
            heads += [("action", "fc_visual_action", act_start, nv)]
            self.reg = [("reg_visual", "fc_visual_action", act_start, nv)] if det else []
        else:
            heads += [("audio", "fc_audio_action", S - na, na)]
            self.reg = [("reg_audio", "fc_audio_action", S - na, na)] if det else []
        for (_, _, s0, n) in heads:
            if s0 < F or s0 + n > S:
                raise ValueError("head slice [%d,%d) outside the query rows [%d,%d): num_v_queries/num_a_queries "
                                 "do not match the time encodings" % (s0, s0 + n, F, S))
        self.heads = heads
        self._table = {}

    def table(self, dev):
        t = self._table.get(dev)
        if t is None:
            t = torch.tensor(self.rows, dtype=torch.int32).reshape(-1, 4).to(dev)
            self._table[dev] = t
        return t


OUT_SLOTS = ("verb", "noun", "action", "audio", "feats", "reg_visual", "reg_audio")


def _gather_head_rows(rt, x, B, S, E, ranges, st, prec=None):
    """ranges: [(s0, n, rows[B*n, E])]: rows = x[b, s0 + i, :] (x: [B*S, E]; element type of precision `prec`, default rt's)"""
    prec = rt.prec if prec is None else prec
    if 2 <= len(ranges) <= 6:
        call("timhip_gather_ranges", prec, ptr(x), B, S, E, len(ranges), _iarr([r[0] for r in ranges]),
             _iarr([r[1] for r in ranges]), _parr([r[2] for r in ranges]), st)
    else:
        for s0, n, rows in ranges:
            call("timhip_gather_rows", prec, ptr(x), B, S, E, s0, n, ptr(rows), st)


class EncoderFn(torch.autograd.Function):
    """forward(ctx, model, nv, na, visual, audio, te, *params) -> 7 outputs (OUT_SLOTS; None if absent).
    `params` is `model._encoder_param_list()` so that autograd tracks every parameter."""

    @staticmethod
    def forward(ctx, model, nv, na, visual, audio, te, *params):
        rt, cfg = model.rt, model.cfg
        _require_gpu(te, "encoder")
        dev = te.device
        P = dict(zip(model._encoder_param_names, params))
        B, T, d = te.shape
        E, FF, H, Lyr, nf = cfg.E, cfg.FF, cfg.nhead, cfg.num_layers, cfg.num_feats
        if d != cfg.d_model:
            raise ValueError("time encodings have width %d, model d_model is %d" % (d, cfg.d_model))
        plan = model._plan(T, nv, na)
        S, F = plan.S, plan.F
        M = B * S
        training = model.training
        if training and rt.recast_every_forward:
            rt.invalidate_weights()
        seed = rt.next_seed() if training else 0
        p_feat = cfg.feat_drop if training else 0.0
        p_seq = cfg.seq_drop if training else 0.0
        p_enc = cfg.enc_dropout if training else 0.0
        te_c = _f32c(te)
        st = _stream()
        fe = "feature_encoding."

        # ---- per-layer saved blocks and, in training, the keep-bits of every layer's attention dropout: ONE launch ahead of the
        # stack (timhip_attn_keep_bits -> the layers' saved blocks; 128-wide heads with 97 .. 128 feature keys: C2a / C3 / C4).  The
        # attention forward and the fused backward then read 16 bytes per row instead of running Philox per (row, key) in both
        # directions.  Same stream of random numbers, same masks (tests/test_gpu_train_parity.py); TIM_AMD_ATTN_KEEP_BITS=0: the
        # kernels draw their own (A/B switch).  TIM_AMD_KEEP_BITS_SIDE=1 (A/B switch): the launch - VALU-bound, 16 us, independent
        # of the front end - on the runtime's side stream under the embedders / the weight refresh, joined in front of layer 0
        desc = L.TimDesc(B, S, F, d, E, H, FF, rt.prec, p_enc, seed, 0, rt.layer_split_flags(E, FF), None)
        saved_bytes = L.load().timhip_layer_saved_bytes(C.byref(desc))
        layer_saved = [torch.empty(saved_bytes, dtype=torch.uint8, device=dev) for _ in range(Lyr)]
        keep_flag = 0
        bits_pending = None
        if p_enc > 0.0 and rt.h16 and os.environ.get("TIM_AMD_ATTN_KEEP_BITS", "1") != "0":
            if os.environ.get("TIM_AMD_KEEP_BITS_SIDE", "0") == "1":
                aux_s = rt.aux_stream(dev)
                aux_s.wait_stream(torch.cuda.current_stream())   # (the blocks' previous owners are done)
                rc = L.load().timhip_attn_keep_bits(C.byref(desc), Lyr, _parr(layer_saved), aux_s.cuda_stream)
                bits_pending = aux_s
            else:
                rc = L.load().timhip_attn_keep_bits(C.byref(desc), Lyr, _parr(layer_saved), st)
            if rc == 0:
                keep_flag = L.DESC_ATTN_KEEP_BITS
                desc.reserved |= keep_flag
            elif rc != L.EUNSUPPORTED:
                L.check(rc, "timhip_attn_keep_bits")

        # ---- modality embedders: e = LN(GELU(drop(x) W^T + b))  (encodings.py:21-26,140-153)
        # Two modalities: their rows are STACKED ([visual | audio], R = B * nf rows each) so that the casts, the projections and the
        # LayerNorms are one launch each (timhip_cast_rows_pair, the grouped GEMM, timhip_layernorm_fwd2; the backward's
        # LayerNorm needs the halves to meet at a multiple of its 16-row blocks)
        emb_saved = []
        emb_gemms = []
        e_bufs = [None, None]
        R = B * nf
        for name, slot in plan.embedders:
            x = visual if name == "visual" else audio
            if x.dim() != 3 or x.shape[0] != B or x.shape[1] != nf:
                raise ValueError("%s input must be [B=%d, num_feats=%d, C], got %s" % (name, B, nf, tuple(x.shape)))
        ne = len(plan.embedders)
        pair = ne == 2 and R % 16 == 0 and os.environ.get("TIM_AMD_EMBEDDER_PAIR", "1") != "0"   # (env: A/B switch)
        u_all = torch.empty((ne * R, d), dtype=torch.float32, device=dev)
        e_all = torch.empty((ne * R, d), dtype=torch.float32, device=dev)
        stats_all = torch.empty((ne * R, 2), dtype=torch.float32, device=dev)
        x2s, xTs, sites = [], [], []
        for i, (name, slot) in enumerate(plan.embedders):
            x = visual if name == "visual" else audio
            Cin = x.shape[2]
            x2s.append(_f32c(x).reshape(R, Cin))
            xTs.append(torch.empty((R, _ru(Cin)), dtype=rt.op_dtype, device=dev))
            sites.append(L.SITE_FEAT_V if name == "visual" else L.SITE_FEAT_A)
        if pair:
            call("timhip_cast_rows_pair", rt.prec, _parr(x2s), _iarr([t.shape[1] for t in x2s]), _parr(xTs),
                 _iarr([t.shape[1] for t in xTs]), R, p_feat, seed, (C.c_uint32 * 2)(*sites), st)
        for i, (name, slot) in enumerate(plan.embedders):
            x2, xT, site = x2s[i], xTs[i], sites[i]
            Cin = x2.shape[1]
            if not pair:
                call("timhip_cast_rows", rt.prec, ptr(x2), R, Cin, Cin, ptr(xT), xT.shape[1], p_feat, seed, site, None, st)
            w = P[fe + name + "_embedder.1.weight"]
            u = u_all[i * R:(i + 1) * R]
            emb_gemms.append(dict(A=xT, B=rt.weight(w), M=R, N=d, K=Cin, out0=u, ld0=d,
                                  bias=_f32c(P[fe + name + "_embedder.1.bias"])))
            emb_saved.append((name, slot, xT, u, stats_all[i * R:(i + 1) * R], Cin, site))
            e_bufs[slot] = e_all[i * R:(i + 1) * R]
        rt.gemm_many(L.EPI_STORE_F32, emb_gemms)   # the two modality embedders (under-filled, independent): one grouped launch
        del emb_gemms, x2s
        lnp = [(_f32c(P[fe + name + "_embedder.3.weight"]), _f32c(P[fe + name + "_embedder.3.bias"])) for name, _ in plan.embedders]
        if pair:
            call("timhip_layernorm_fwd2", rt.prec, ptr(u_all), 2 * R, d, d, 2, ptr(lnp[0][0]), ptr(lnp[0][1]), R, ptr(lnp[1][0]),
                 ptr(lnp[1][1]), ptr(e_all), d, None, 0, ptr(stats_all), st)
        else:
            for i, (name, slot, xT, u, stats, Cin, site) in enumerate(emb_saved):
                rt.ln_fwd(u, R, d, 2, lnp[i][0], lnp[i][1], xf=e_bufs[slot], ldx=d, stats=stats)

        # ---- sequence assembly (encodings.py:190-250)
        # (the CLS token / modality vectors go to the kernel by pointer: separate parameters, no concatenation launches)
        cls_v = [_f32c(P[fe + n]).reshape(-1) for n in plan.cls_names]
        mod_v = [_f32c(P[fe + n]).reshape(-1) for n in plan.mod_names]
        # fp32 rows exist only at the two ends of the stack: the assembled input and the last layer's output (-> feats);
        # between layers every reader normalises the previous layer's pre-norm rows itself (timhip_layer_fwd_chained)
        xs_f = [None] * (Lyr + 1)
        xs_f[0] = torch.empty((M, E), dtype=torch.float32, device=dev)
        xs_f[Lyr] = torch.empty((M, E), dtype=torch.float32, device=dev)
        xs_t = [torch.empty((M, E), dtype=rt.op_dtype, device=dev) for _ in range(Lyr + 1)]
        tab = plan.table(dev)
        call("timhip_assemble_fwd_p", rt.prec, ptr(tab), B, S, d, ptr(e_bufs[0]), ptr(e_bufs[1]), nf, _parr(cls_v), len(cls_v),
             ptr(te_c), T, _parr(mod_v), len(mod_v), p_seq, seed, L.SITE_SEQ, ptr(xs_f[0]), ptr(xs_t[0]), st)

        # ---- L post-norm encoder layers (transformers.py:44-45,92-111)
        ws_bytes = L.load().timhip_layer_workspace_bytes(C.byref(desc))
        ws = model._workspace(ws_bytes, dev)
        stack = model._stack_prefix
        if bits_pending is not None:   # the keep-bits launch of the side stream joins here
            torch.cuda.current_stream().wait_stream(bits_pending)
        lparams = []
        for l in range(Lyr):
            pre = "%s.layers.%d." % (stack, l)
            lp = model._layer_params(rt, P, pre)
            lparams.append(lp)
            sv = layer_saved[l]
            desc.layer = l
            if l == 0:
                call("timhip_layer_fwd", C.byref(desc), C.byref(lp[0]), ptr(xs_f[0]), ptr(xs_t[0]), ptr(xs_f[1]),
                     ptr(xs_t[1]), ptr(sv), ptr(ws), ws_bytes, st)
            else:
                call("timhip_layer_fwd_chained", C.byref(desc), C.byref(lp[0]), C.byref(lparams[l - 1][0]),
                     ptr(layer_saved[l - 1]), ptr(xs_t[l]), ptr(xs_f[l + 1]), ptr(xs_t[l + 1]), ptr(sv), st)

        # ---- heads (head.py:17-38).  fp16 model: the logits are produced with split operands from the fp32 rows of the last
        # layer; the backward gathers the fp16 rows it needs itself
        xL_t = xs_t[Lyr]
        outs = {}
        head_saved = []
        head_gemms, head_ranges, head_splits = [], [], []
        for slot, pname, s0, n in plan.heads:
            w = P["cls_head." + pname + ".weight"]
            Cn = w.shape[0]
            logits = torch.empty((B * n, Cn), dtype=torch.float32, device=dev)
            bias = _f32c(P["cls_head." + pname + ".bias"])
            if rt.split:   # fp32 rows of the last layer -> [hi | lo | hi] fp16 blocks, weights [hi | hi | lo]: K = 3 E
                rows3 = torch.empty((B * n, 3 * E), dtype=rt.op_dtype, device=dev)
                if n > 0:
                    head_ranges.append((s0, n, rows3))
                    head_gemms.append(dict(A=rows3, B=rt.weight_split(w), M=B * n, N=Cn, K=3 * E, out0=logits, ld0=Cn, bias=bias, rep=3))
                # (the backward's fp16 rows ARE the first block of rows3: T(fp32 row) = the operand copy the last LayerNorm wrote)
                head_saved.append((slot, pname, s0, n, rows3[:, :E] if n > 0 else None))
            else:
                rows = torch.empty((B * n, E), dtype=rt.op_dtype, device=dev)
                if n > 0:
                    head_ranges.append((s0, n, rows))
                    head_gemms.append(dict(A=rows, B=rt.weight(w), M=B * n, N=Cn, K=E, out0=logits, ld0=Cn, bias=bias))
                head_saved.append((slot, pname, s0, n, rows))
            outs[slot] = logits
        # the heads' row gathers and GEMMs are independent and tiny: one launch of each kind for all of them
        if rt.split:   # gathered and split in one launch (E is a multiple of 64: d_model % 32 == 0)
            for i0 in range(0, len(head_ranges), 6):
                grp = head_ranges[i0:i0 + 6]
                call("timhip_gather_split3_ranges", rt.prec, ptr(xs_f[Lyr]), B, S, E, len(grp), _iarr([r[0] for r in grp]),
                     _iarr([r[1] for r in grp]), _parr([r[2] for r in grp]), st)
        else:
            _gather_head_rows(rt, xL_t, B, S, E, head_ranges, st)
        rt.gemm_many(L.EPI_STORE_F32, head_gemms)
        del head_gemms, head_ranges, head_splits
        reg_saved = []
        for slot, pname, s0, n in plan.reg:
            pre = "reg_head." + pname + "."
            hid = E // 2
            rows = torch.empty((B * n, E), dtype=rt.op_dtype, device=dev)
            h1 = rt.out_op(B * n, hid, dev)    # (zero-filled only when hid is not a multiple of 64: padding columns)
            h2 = rt.out_op(B * n, hid, dev)
            y = torch.empty((B * n, 2), dtype=torch.float32, device=dev)
            if n > 0:
                call("timhip_gather_rows", rt.prec, ptr(xL_t), B, S, E, s0, n, ptr(rows), st)
                rt.gemm(L.EPI_RELU_T, rows, rt.weight(P[pre + "0.weight"]), B * n, hid, E, h1, h1.shape[1],
                        bias=_f32c(P[pre + "0.bias"]))
                rt.gemm(L.EPI_RELU_T, h1, rt.weight(P[pre + "2.weight"]), B * n, hid, hid, h2, h2.shape[1],
                        bias=_f32c(P[pre + "2.bias"]))
                rt.gemm(L.EPI_SIGMOID_F32, h2, rt.weight(P[pre + "4.weight"]), B * n, 2, hid, y, 2,
                        bias=_f32c(P[pre + "4.bias"]))
            outs[slot] = y
            reg_saved.append((slot, pname, s0, n, rows, h1, h2, y))
        feats = xs_f[Lyr].view(B, S, E)[:, :F]
        outs["feats"] = feats

        ctx.model, ctx.plan, ctx.P_names = model, plan, model._encoder_param_names
        ctx.dims = (B, T, d, S, F, M, nv, na)
        ctx.drop = (p_feat, p_seq, p_enc, seed)
        ctx.keep_flag = keep_flag
        ctx.salt_epoch = _SALT["epoch"] if training else None
        ctx.emb_saved, ctx.layer_saved, ctx.head_saved, ctx.reg_saved = emb_saved, layer_saved, head_saved, reg_saved
        ctx.emb_pair = (u_all, stats_all) if pair else None
        ctx.xs_t, ctx.lparams = xs_t, lparams
        ctx.in_shapes = (tuple(visual.shape), tuple(audio.shape))
        ctx.save_for_backward(*params)
        result = tuple(outs.get(k) for k in OUT_SLOTS)
        ctx.mark_non_differentiable(*[])
        return result

    @staticmethod
    def backward(ctx, *gouts):
        model, plan = ctx.model, ctx.plan
        rt, cfg = model.rt, model.cfg
        params = ctx.saved_tensors
        names = ctx.P_names
        P = dict(zip(names, params))
        B, T, d, S, F, M, nv, na = ctx.dims
        p_feat, p_seq, p_enc, seed = ctx.drop
        if _SALT["word"] is not None and ctx.salt_epoch is not None and ctx.salt_epoch != _SALT["epoch"]:
            raise RuntimeError("graph-safe dropout: another training forward ran between this forward and its backward; "
                               "the masks are regenerated from the device-side salt, so each forward needs its backward first")
        E, FF, H, Lyr, nf = cfg.E, cfg.FF, cfg.nhead, cfg.num_layers, cfg.num_feats
        dev = params[0].device
        st = _stream()
        g = dict(zip(OUT_SLOTS, gouts))
        fe = "feature_encoding."

        # gradient buckets: one flat fp32 buffer per bucket, parameters are views into it
        # bf16: the layers' Linear gradients are written, not accumulated -> their buckets are not zero-filled (nor read)
        overwrite = rt.h16
        # every buffer of this pass that has to start at zero goes out in ONE multi-tensor launch with the buckets' fills: the
        # gradient-scale block of the fp16 mode (the cls / modality gradient sums are accumulated straight into their views of
        # the zero-filled front-end bucket)
        gs_block = torch.empty(8, dtype=torch.float32, device=dev) if rt.prec == L.PREC_F16 else None
        extra_zero = [gs_block] if gs_block is not None else []
        grads = model._alloc_grad_buckets(names, params, dev, layer_overwrite=overwrite, extra_zero=extra_zero)
        G = grads.views

        # gradient stream of the last layer's output: the feature rows start as the incoming `feats` cotangent (a copy, not a
        # zero fill + add: 66 instead of 118 MB at C2a), the query rows as zeros for the heads' row scatters to add into
        dx = torch.empty((M, E), dtype=torch.float32, device=dev)   # (written below, behind the heads' input-gradient products)
        dx3 = dx.view(B, S, E)
        xL_t = ctx.xs_t[Lyr]
        # fp16: scale of the gradient operands for this pass, from the cotangents that enter it (device side, no sync).  The
        # fp32 stream dx and every parameter gradient stay true-scale; only fp16 tensors carry the factor.
        gs = rt.grad_scale([_f32c(v) for v in gouts if v is not None], dev, out=gs_block)
        gs_in, gs_out = (ptr(gs), ptr(gs) + 4) if gs is not None else (None, None)
        # (fp16: the forward fed the heads from the fp32 rows as [hi | lo | hi] blocks; the weight gradients read the hi block -
        #  row stride 3 E - as their fp16 activations: no second gather)
        head_saved = ctx.head_saved

        # ---- heads (their weight gradients are collected and launched grouped by row count)
        wg_items = []
        head_dgrads, head_scatter, head_casts = [], [], []
        for slot, pname, s0, n, rows in head_saved:
            go = g[slot]
            if go is None or n == 0:
                continue
            w = P["cls_head." + pname + ".weight"]
            Cn = w.shape[0]
            go = _f32c(go)
            gT = torch.empty((B * n, _ru(Cn)), dtype=rt.op_dtype, device=dev)
            head_casts.append((go, B * n, Cn, gT))
            wg_items.append((gT, Cn, rows, E, B * n, G["cls_head." + pname + ".weight"], G["cls_head." + pname + ".bias"]))
            # A head with a long contraction (the 3806 action classes) as column chunks of the contraction side by side, each into
            # its own fp32 slab (dx_init adds them up): as ONE item its 240 blocks ran 60 contraction steps while the other heads'
            # blocks had finished after 2 - 5 (49 us for 8 GFLOP).  At most six items per grouped launch.
            Kp = _ru(Cn)
            ns = 1
            if rt.h16 and Kp >= 2048:
                ns = max(1, min(4, Kp // 1024, 6 - (len(plan.heads) - 1)))
            wT = rt.weight(w, True)
            if ns == 1:
                d_rows = torch.empty((B * n, E), dtype=torch.float32, device=dev)
                head_dgrads.append(dict(A=gT, B=wT, M=B * n, N=E, K=Cn, out0=d_rows, ld0=E))
            else:
                d_rows = torch.empty((ns, B * n, E), dtype=torch.float32, device=dev)
                step = _ru((Kp + ns - 1) // ns)
                for j in range(ns):
                    k0, k1 = j * step, min(Kp, (j + 1) * step)
                    head_dgrads.append(dict(A=gT[:, k0:k1], B=wT[:, k0:k1], M=B * n, N=E, K=k1 - k0, out0=d_rows[j], ld0=E))
            head_scatter.append((d_rows, s0, n, ns))
        # cotangent casts, input-gradient GEMMs and row scatters of all heads: one launch of each kind
        if 2 <= len(head_casts) <= 6:
            call("timhip_cast_rows_many", rt.prec, len(head_casts), _parr([c[0] for c in head_casts]),
                 _iarr([c[1] for c in head_casts]), _iarr([c[2] for c in head_casts]), _parr([c[3] for c in head_casts]),
                 _iarr([c[3].shape[1] for c in head_casts]), gs_in, st)
        else:
            for go, r_, c_, gT in head_casts:
                call("timhip_cast_rows", rt.prec, ptr(go), r_, c_, c_, ptr(gT), gT.shape[1], 0.0, 0, 0, gs_in, st)
        rt.gemm_many(L.EPI_ADD_F32, head_dgrads, acc_scale=gs_out)
        spans = sorted((h[1], h[1] + h[2]) for h in head_scatter)
        disjoint = all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))   # detection: several heads read one row
        gfeats = _f32c(g["feats"]) if g["feats"] is not None else None
        if len(head_scatter) <= 6 and disjoint:
            # one pass writes the whole stream: feature rows <- the `feats` cotangent, query rows <- their head's rows, rest 0
            call("timhip_dx_init_slabs", B, S, F, E, ptr(gfeats), len(head_scatter), _iarr([h[1] for h in head_scatter]),
                 _iarr([h[2] for h in head_scatter]), _parr([h[0] for h in head_scatter]), _iarr([h[3] for h in head_scatter]),
                 ptr(dx), st)
        else:
            if gfeats is not None:
                dx3[:, :F].copy_(gfeats)
            else:
                dx3[:, :F].zero_()
            if S > F:
                dx3[:, F:].zero_()
            for d_rows, s0, n, ns in head_scatter:
                for j in range(ns):
                    call("timhip_scatter_rows_add", ptr(d_rows[j] if ns > 1 else d_rows), B, S, E, s0, n, ptr(dx), st)
        del head_dgrads, head_scatter, head_casts
        for slot, pname, s0, n, rows, h1, h2, y in ctx.reg_saved:
            go = g[slot]
            if go is None or n == 0:
                continue
            pre = "reg_head." + pname + "."
            hid = E // 2
            # sigmoid backward on the [B*n, 2] outputs, written as the zero-padded operand rows of the gradient GEMMs
            gzT = torch.empty((B * n, 64), dtype=rt.op_dtype, device=dev)
            call("timhip_sigmoid_bwd_rows", rt.prec, ptr(_f32c(go)), ptr(y), B * n, 2, ptr(gzT), 64, gs_in, st)
            wg_items.append((gzT, 2, h2, hid, B * n, G[pre + "4.weight"], G[pre + "4.bias"]))
            dh2 = rt.out_op(B * n, hid, dev)
            rt.gemm(L.EPI_DRELU_T, gzT, rt.weight(P[pre + "4.weight"], True), B * n, hid, 2, dh2, dh2.shape[1],
                    aux=h2, ldaux=h2.shape[1])
            wg_items.append((dh2, hid, h1, hid, B * n, G[pre + "2.weight"], G[pre + "2.bias"]))
            dh1 = rt.out_op(B * n, hid, dev)
            rt.gemm(L.EPI_DRELU_T, dh2, rt.weight(P[pre + "2.weight"], True), B * n, hid, hid, dh1, dh1.shape[1],
                    aux=h1, ldaux=h1.shape[1])
            wg_items.append((dh1, hid, rows, E, B * n, G[pre + "0.weight"], G[pre + "0.bias"]))
            d_rows = torch.empty((B * n, E), dtype=torch.float32, device=dev)
            rt.gemm(L.EPI_ADD_F32, dh1, rt.weight(P[pre + "0.weight"], True), B * n, E, hid, d_rows, E, acc_scale=gs_out)
            call("timhip_scatter_rows_add", ptr(d_rows), B, S, E, s0, n, ptr(dx), st)
        rt.wgrad_many(wg_items, out_scale=gs_out)
        del wg_items
        grads.done("heads")

        # ---- layers, last to first.  Per layer: the data chain on the current stream, the four
        # weight-gradient GEMMs on a side stream (they overlap the data chain of the next layer);
        # each layer's bucket is handed to the hook as soon as its weight gradients are enqueued.
        desc = L.TimDesc(B, S, F, d, E, H, FF, rt.prec, p_enc, seed, 0,
                         (L.DESC_WGRAD_OVERWRITE if overwrite else 0) | (L.DESC_WGRAD_SEPARATE if rt.separate_wgrad else 0) | ctx.keep_flag,
                         gs_in)
        lib = L.load()
        dx2 = torch.empty_like(dx)
        # between layers the gradient travels split: fp32 part (what LayerNorm-backward wrote) + operand-dtype part (the
        # in-projection's input-gradient product, added by the next LayerNorm-backward as it reads): timhip_layer_bwd_split
        # (plain bf16 would round that part to 8 bits per layer: there the layer returns one complete fp32 gradient instead)
        split_stream = Lyr > 1 and rt.prec != L.PREC_BF16
        dxa = [torch.empty((M, E), dtype=rt.op_dtype, device=dev) for _ in range(2)] if split_stream else [None, None]
        # fp16, TIM_AMD_GRAD_STREAM=16 (opt-in): the residual part travels 16-bit as well, under the same gradient scale
        # (TIMHIP_DESC_STREAM16*): the stack's entry (dx_init) and exit (layer 0 -> assemble_bwd) stay fp32.
        stream16 = split_stream and rt.prec == L.PREC_F16 and gs is not None and rt.grad_stream16
        dxh = [torch.empty((M, E), dtype=rt.op_dtype, device=dev) for _ in range(2)] if stream16 else [None, None]
        base_flags = desc.reserved
        add_in = None     # 16-bit part of the gradient entering the current layer (None at the top of the stack)
        stack = model._stack_prefix
        main = torch.cuda.current_stream()
        overlap = rt.overlap_wgrad
        keep_alive = []
        if overlap:
            aux = rt.aux_stream(dev)
            dws_bytes = lib.timhip_layer_data_workspace_bytes(C.byref(desc))
            wws_bytes = lib.timhip_layer_wgrad_workspace_bytes(C.byref(desc))
            dy_bytes = lib.timhip_layer_dy_bytes(C.byref(desc))
            ws = model._workspace(dws_bytes, dev)
            wws = model._workspace(wws_bytes, dev, slot="wgrad")
            dys = [model._workspace(dy_bytes, dev, slot="dy0"), model._workspace(dy_bytes, dev, slot="dy1")]
            done = {}
            aux.wait_stream(main)  # gradient buckets were zeroed on the main stream
        else:
            ws_bytes = lib.timhip_layer_workspace_bytes(C.byref(desc))
            ws = model._workspace(ws_bytes, dev)
        # round 6: where two layers' weight gradients are ONE round of eight-phase tiles (timhip_layer_wgrad_pair_wins: C2a at
        # production batch sizes), a layer's weight gradients wait for its neighbour's data chain and the pair goes out as one
        # launch on the same stream; each layer of a pair keeps its own `dy` block (TIM_AMD_WGRAD_PAIR=0: A/B switch)
        pair = (not overlap and Lyr >= 2 and os.environ.get("TIM_AMD_WGRAD_PAIR", "1") != "0"
                and lib.timhip_layer_wgrad_pair_wins(C.byref(desc)) == 1)
        pending = None
        if pair:
            dws_bytes = lib.timhip_layer_data_workspace_bytes(C.byref(desc))
            wws_bytes = lib.timhip_layer_wgrad_workspace_bytes(C.byref(desc))
            dy_bytes = lib.timhip_layer_dy_bytes(C.byref(desc))
            ws = model._workspace(dws_bytes, dev)
            wws = model._workspace(wws_bytes, dev, slot="wgrad")
            dys = [model._workspace(dy_bytes, dev, slot="dy0"), model._workspace(dy_bytes, dev, slot="dy1")]
        # LayerNorm dgamma / dbeta: every layer leaves per-block partials, one launch reduces them all at the end - unless a
        # data-parallel hook takes each layer's bucket as soon as the layer is done (then the layer call reduces its own)
        defer_ln = rt.bucket_hook is None and os.environ.get("TIM_AMD_NO_DEFER_LN", "0") != "1"   # (env: A/B switch)
        ln_part_bytes = lib.timhip_layer_ln_partial_bytes(C.byref(desc)) if defer_ln else 0
        ln_part = torch.empty(Lyr * ln_part_bytes, dtype=torch.uint8, device=dev) if defer_ln else None
        def _stream16_io(l):
            """(gradient entering layer l, gradient leaving it) and the layer's stream flags: fp32 `dx` / `dx2` at the two ends of
            the stack, the 16-bit ping-pong pair in between"""
            if not stream16:
                desc.reserved = base_flags
                return dx, dx2
            top, bottom = l == Lyr - 1, l == 0
            desc.reserved = base_flags | L.DESC_STREAM16 | (0 if top else L.DESC_STREAM16_IN) | (0 if bottom else L.DESC_STREAM16_OUT)
            return (dx if top else dxh[(l + 1) & 1]), (dx2 if bottom else dxh[l & 1])

        for l in reversed(range(Lyr)):
            pre = "%s.layers.%d." % (stack, l)
            lg = L.TimLayerGrads(*[ptr(G[pre + n]) for n in model._LAYER_GRAD_NAMES])
            if defer_ln:
                lg.ln_partials = ptr(ln_part) + l * ln_part_bytes
            desc.layer = l
            if overlap:
                dyb = dys[l & 1]
                if l + 2 in done:
                    main.wait_event(done[l + 2])  # the weight gradients of layer l+2 no longer read this dy
                add_out = dxa[l & 1] if l > 0 else None
                s_in, s_out = _stream16_io(l)
                call("timhip_layer_bwd_data_split", C.byref(desc), C.byref(ctx.lparams[l][0]), ptr(ctx.layer_saved[l]),
                     ptr(s_in), ptr(add_in), ptr(s_out), ptr(add_out), ptr(dyb), C.byref(lg), ptr(ws), dws_bytes, main.cuda_stream)
                add_in = add_out
                ev = torch.cuda.Event()
                ev.record(main)
                aux.wait_event(ev)
                call("timhip_layer_bwd_weights", C.byref(desc), ptr(ctx.xs_t[l]), ptr(ctx.layer_saved[l]), ptr(dyb),
                     C.byref(lg), ptr(wws), wws_bytes, aux.cuda_stream)
                dn = torch.cuda.Event()
                dn.record(aux)
                done[l] = dn
                keep_alive.append(ctx.layer_saved[l])
                grads.done("layer%d" % l, ready=dn)
            elif pair:
                dyb = dys[l & 1]
                add_out = dxa[l & 1] if l > 0 else None
                s_in, s_out = _stream16_io(l)
                call("timhip_layer_bwd_data_split", C.byref(desc), C.byref(ctx.lparams[l][0]), ptr(ctx.layer_saved[l]),
                     ptr(s_in), ptr(add_in), ptr(s_out), ptr(add_out), ptr(dyb), C.byref(lg), ptr(ws), dws_bytes, st)
                add_in = add_out
                if pending is None and l > 0:
                    pending = (l, lg)      # its weight gradients go out with layer l - 1's
                elif pending is not None:
                    lp, lgp = pending
                    call("timhip_layer_bwd_weights_pair", C.byref(desc), ptr(ctx.xs_t[lp]), ptr(ctx.layer_saved[lp]), ptr(dys[lp & 1]),
                         C.byref(lgp), ptr(ctx.xs_t[l]), ptr(ctx.layer_saved[l]), ptr(dyb), C.byref(lg), ptr(wws), wws_bytes, st)
                    grads.done("layer%d" % lp)
                    grads.done("layer%d" % l)
                    ctx.layer_saved[lp] = None
                    pending = None
                else:                      # an odd layer count: layer 0 on its own
                    call("timhip_layer_bwd_weights", C.byref(desc), ptr(ctx.xs_t[l]), ptr(ctx.layer_saved[l]), ptr(dyb),
                         C.byref(lg), ptr(wws), wws_bytes, st)
                    grads.done("layer%d" % l)
            else:
                add_out = dxa[l & 1] if l > 0 else None
                s_in, s_out = _stream16_io(l)
                call("timhip_layer_bwd_split", C.byref(desc), C.byref(ctx.lparams[l][0]), ptr(ctx.xs_t[l]),
                     ptr(ctx.layer_saved[l]), ptr(s_in), ptr(add_in), ptr(s_out), ptr(add_out), C.byref(lg), ptr(ws), ws_bytes, st)
                add_in = add_out
                grads.done("layer%d" % l)
            dx, dx2 = dx2, dx
            if pending is None or pending[0] != l:
                ctx.layer_saved[l] = None

        # LayerNorm parameter gradients of all layers: one reduction of the saved per-block partials (sets: norm2, norm1 per layer)
        dgs, dbs = [], []
        for l in range(Lyr if defer_ln else 0):
            pre = "%s.layers.%d." % (stack, l)
            dgs += [G[pre + "norm2.weight"], G[pre + "norm1.weight"]]
            dbs += [G[pre + "norm2.bias"], G[pre + "norm1.bias"]]
        for i0 in range(0, len(dgs), 16):
            call("timhip_ln_partials_reduce", ptr(ln_part) + (i0 // 2) * ln_part_bytes, len(dgs[i0:i0 + 16]), M, E,
                 _parr(dgs[i0:i0 + 16]), _parr(dbs[i0:i0 + 16]), st)
        del ln_part

        # ---- sequence assembly backward
        d_e = [None, None]
        d_e_all = torch.empty((len(ctx.emb_saved) * B * nf, d), dtype=torch.float32, device=dev)   # (stacked like the forward's rows)
        for i, (name, slot, *_) in enumerate(ctx.emb_saved):
            d_e[slot] = d_e_all[i * B * nf:(i + 1) * B * nf]
        d_te = torch.empty((B, T, d), dtype=torch.float32, device=dev)   # written in full by the kernel
        # cls / modality gradients: atomics accumulate straight into the parameters' views of the (zero-filled) front-end bucket
        d_cls = [G[fe + n].view(-1) for n in plan.cls_names]
        d_mod = [G[fe + n].view(-1) for n in plan.mod_names]
        call("timhip_assemble_bwd_p", ptr(plan.table(dev)), B, S, d, ptr(dx), nf, T, p_seq, seed, L.SITE_SEQ,
             ptr(d_e[0]), ptr(d_e[1]), _parr(d_cls), len(d_cls), ptr(d_te), _parr(d_mod), len(d_mod), st)

        # ---- embedders backward
        d_inputs = {"visual": None, "audio": None}
        emb_items = []
        need_in = {"visual": ctx.needs_input_grad[3], "audio": ctx.needs_input_grad[4]}
        R = B * nf
        duT_all = rt.out_op(len(ctx.emb_saved) * R, d, dev)
        if ctx.emb_pair is not None:   # both modalities' LayerNorm backward as one launch over the stacked rows
            (n0, _, _, _, _, _, _), (n1, _, _, _, _, _, _) = ctx.emb_saved
            u_all, stats_all = ctx.emb_pair
            call("timhip_layernorm_bwd2", rt.prec, ptr(d_e_all), d, ptr(u_all), d, ptr(stats_all), 2 * R, d, 2,
                 ptr(_f32c(P[fe + n0 + "_embedder.3.weight"])), R, ptr(_f32c(P[fe + n1 + "_embedder.3.weight"])), None, 0,
                 ptr(duT_all), duT_all.stride(0), ptr(G[fe + n0 + "_embedder.3.weight"]), ptr(G[fe + n0 + "_embedder.3.bias"]),
                 ptr(G[fe + n1 + "_embedder.3.weight"]), ptr(G[fe + n1 + "_embedder.3.bias"]), gs_in, st)
        for i, (name, slot, xT, u, stats, Cin, site) in enumerate(ctx.emb_saved):
            w = P[fe + name + "_embedder.1.weight"]
            duT = duT_all[i * R:(i + 1) * R]
            if ctx.emb_pair is None:
                rt.ln_bwd(d_e[slot], u, stats, R, d, 2, _f32c(P[fe + name + "_embedder.3.weight"]), dyt=duT,
                          dgamma=G[fe + name + "_embedder.3.weight"], dbeta=G[fe + name + "_embedder.3.bias"], t_scale=gs_in)
            emb_items.append((duT, d, xT, Cin, R, G[fe + name + "_embedder.1.weight"], G[fe + name + "_embedder.1.bias"]))
            if need_in[name]:
                gx = torch.empty((R, Cin), dtype=torch.float32, device=dev)
                rt.gemm(L.EPI_ADD_F32, duT, rt.weight(w, True), R, Cin, d, gx, Cin, acc_scale=gs_out)
                dxin = torch.empty((R, Cin), dtype=torch.float32, device=dev)
                call("timhip_dropout_rows_bwd", ptr(gx), R, Cin, Cin, ptr(dxin), Cin, p_feat, seed, site, st)
                d_inputs[name] = dxin.view(B, nf, Cin)
        rt.wgrad_many(emb_items, out_scale=gs_out)
        del emb_items
        grads.done("front")
        if overlap:
            main.wait_stream(aux)  # all weight gradients are complete before autograd sees them
        del keep_alive
        if rt.finish_hook is not None:
            rt.finish_hook()
        out = [None, None, None, d_inputs["visual"], d_inputs["audio"], d_te if ctx.needs_input_grad[5] else None]
        out += [G[n] if ctx.needs_input_grad[6 + i] else None for i, n in enumerate(names)]
        return tuple(out)
