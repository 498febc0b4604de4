"""`TIM`: drop-in for the reference's `time_interval_machine.models.tim.TIM`
(recognition/time_interval_machine/models/tim.py:17-191 and the detection variant
detection/time_interval_machine/models/tim.py:17-430).

Same constructor arguments, same `forward(inputs, forward_type, ...)` dispatch, same
parameter names and shapes (state_dict round-trips with reference checkpoints), same
output tuple.  The arithmetic of `"time_mlp"` and `"encoder"` runs in the HIP library
(`tim_amd.functional`); the torch.nn modules below are parameter containers only and
their own `forward` is never called on that path.
"""
import copy
import math

import torch
from torch import nn
from torch.nn.init import normal_

from . import _lib as L
from .config import TimConfig
from .functional import EncoderFn, EncoderPlan, Runtime, TimeMlpFn, OUT_SLOTS


# ---- parameter containers, registered in the reference's order so that state_dict() key order matches ----
class _FeatureEncodingParams(nn.Module):
    """encodings.py:7-39, 77-100, 123-179 (rec) / 7-33, 55-81, 102-150 (det)."""

    def __init__(self, cfg: TimConfig):
        super().__init__()
        d = cfg.d_model
        det = cfg.variant == "detection"

        def embedder(cin):
            return nn.Sequential(nn.Dropout(p=cfg.feat_drop), nn.Linear(cin, d), nn.GELU(), nn.LayerNorm(d))

        def cls():
            p = nn.Parameter(torch.empty((1, 1, d)))
            normal_(p, std=0.01)
            return p

        if cfg.input_modality == "audio_visual":
            self.visual_embedder = embedder(cfg.visual_input_dim)
            self.audio_embedder = embedder(cfg.audio_input_dim)
            self.visual_modality_encoding = nn.Parameter(torch.empty((1, 1, 2 * d)))
            self.audio_modality_encoding = nn.Parameter(torch.empty((1, 1, 2 * d)))
            normal_(self.visual_modality_encoding, std=0.01)
            normal_(self.audio_modality_encoding, std=0.01)
            if "visual" in cfg.data_modality:
                self.visual_action_cls = cls()
                if cfg.include_verb_noun and not det:
                    self.visual_verb_cls = cls()
                    self.visual_noun_cls = cls()
            if "audio" in cfg.data_modality:
                self.audio_action_cls = cls()
        elif cfg.input_modality == "visual":
            self.visual_embedder = embedder(cfg.visual_input_dim)
            if det:
                self.visual_action_cls = cls()
            else:
                self.action_cls = cls()
                if cfg.include_verb_noun:
                    self.verb_cls = cls()
                    self.noun_cls = cls()
        else:
            self.audio_embedder = embedder(cfg.audio_input_dim)
            if det:
                self.audio_action_cls = cls()
            else:
                self.action_cls = cls()
        self.dropout = nn.Dropout(p=cfg.seq_drop)


class _ClsHeadParams(nn.Module):
    """head.py:4-15,40-51,71-74 (rec) / 7-25,48-63,81-87 (det, with the focal prior bias)."""

    def __init__(self, cfg: TimConfig):
        super().__init__()
        E = cfg.E
        det = cfg.variant == "detection"
        nc = cfg.num_class
        bias_value = -(math.log((1 - 0.01) / 0.01))
        made = []

        def fc(name, n):
            setattr(self, name, nn.Linear(E, n))
            made.append(name)

        if cfg.data_modality == "audio_visual":
            vn = isinstance(nc, list) if det else isinstance(nc[0], list)
            if vn:
                fc("fc_visual_verb", nc[0][0]); fc("fc_visual_noun", nc[0][1]); fc("fc_visual_action", nc[0][2])
            else:
                fc("fc_visual_action", nc[0])
            fc("fc_audio_action", nc[1])
        elif cfg.data_modality == "visual":
            v = nc[0]
            if isinstance(v, list):
                fc("fc_visual_verb", v[0]); fc("fc_visual_noun", v[1]); fc("fc_visual_action", v[2])
            else:
                fc("fc_visual_action", v)
        else:
            fc("fc_audio_action", nc[1])
        if det:
            for n in made:
                nn.init.constant_(getattr(self, n).bias, bias_value)


class _RegHeadParams(nn.Module):
    """det head.py:95-163."""

    def __init__(self, cfg: TimConfig):
        super().__init__()
        E = cfg.E

        def mlp():
            return nn.Sequential(nn.Linear(E, E // 2), nn.ReLU(), nn.Linear(E // 2, E // 2), nn.ReLU(),
                                 nn.Linear(E // 2, 2), nn.Sigmoid())

        if cfg.data_modality in ("audio_visual", "visual"):
            self.fc_visual_action = mlp()
        if cfg.data_modality in ("audio_visual", "audio"):
            self.fc_audio_action = mlp()


class _EncoderLayerParams(nn.Module):
    """transformers.py:71-87."""

    def __init__(self, E, H, FF, p):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(E, H, dropout=p)
        self.dropout1 = nn.Dropout(p)
        self.norm1 = nn.LayerNorm(E)
        self.linear1 = nn.Linear(E, FF)
        self.dropout = nn.Dropout(p)
        self.linear2 = nn.Linear(FF, E)
        self.dropout2 = nn.Dropout(p)
        self.norm2 = nn.LayerNorm(E)


class _EncoderStackParams(nn.Module):
    """transformers.py:27-30: `_get_clones` deep-copies one layer, so a freshly built model
    starts with identical layers, exactly as the reference does."""

    def __init__(self, layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])
        self.num_layers = num_layers


class _AVGAParams(nn.Module):
    """Audio-guided visual attention pooling of the AVE recipe (pool.py:6-43; `pool_features=True`): collapses a [B, T, 7, 7,
    Cv] visual feature map to [B, T, Cv] before the hot path starts.  An input pre-step outside the encoder path (SURVEY 2,
    row 9): parameter names / shapes / initialisation as the reference's so that checkpoints round-trip; evaluated with stock
    torch ops (autograd included)."""

    def __init__(self, a_dim, v_dim, hidden_size, map_size=49):
        super().__init__()
        self.affine_audio = nn.Linear(a_dim, hidden_size)
        self.affine_video = nn.Linear(v_dim, hidden_size)
        self.affine_v = nn.Linear(hidden_size, map_size, bias=False)
        self.affine_g = nn.Linear(hidden_size, map_size, bias=False)
        self.affine_h = nn.Linear(map_size, 1, bias=False)
        for lin in (self.affine_v, self.affine_g, self.affine_h, self.affine_audio, self.affine_video):
            nn.init.xavier_uniform_(lin.weight)
        nn.init.constant_(self.affine_audio.bias, 0)
        nn.init.constant_(self.affine_video.bias, 0)

    def forward(self, audio, video):
        B, T, C = video.shape[0], video.shape[1], video.shape[-1]
        cells = video.reshape(B * T, -1, C)                                   # [B*T, 49, Cv] spatial cells of one time step
        hv = torch.relu(self.affine_video(cells))
        ha = torch.relu(self.affine_audio(audio.reshape(B * T, -1)))
        score = self.affine_h(torch.tanh(self.affine_v(hv) + self.affine_g(ha).unsqueeze(2))).squeeze(2)
        attn = torch.softmax(score, dim=-1)                                   # one weight per spatial cell
        return torch.einsum("rs,rsc->rc", attn, cells).reshape(B, T, C)


class _BucketLayout:
    """Where every parameter's gradient lives inside ONE flat allocation - computed once per (parameter names, shapes, overwrite
    mode) and reused by every backward pass (round 5: building it per pass cost 0.5 ms of host time per C2a step, a sixth of the
    eager step's issue time).  Buckets in the order the backward completes them (heads, last layer ... first layer, front end):
    consecutive buckets are contiguous, so the data-parallel exchange can take several as one range; 256-byte aligned bucket
    starts (the tail padding is exchanged too)."""

    def __init__(self, names, params, bucket_of, layer_overwrite):
        sizes = {}
        for n, p in zip(names, params):
            b = bucket_of(n)
            sizes[b] = sizes.get(b, 0) + (p.numel() + 3) // 4 * 4

        def order(b):
            return (0, 0) if b == "heads" else ((1, -int(b[5:])) if b.startswith("layer") else (2, 0))
        self.order = sorted(sizes, key=order)
        self.bucket = {}                 # bucket -> (start, padded size)
        pos = 0
        for b in self.order:
            n = (sizes[b] + 63) // 64 * 64
            self.bucket[b] = (pos, n)
            pos += n
        self.total = pos
        # zero fills: whole buckets the kernels accumulate into (heads, front end), and of the overwritten layer buckets only
        # the pieces nobody writes (alignment tails) or that are accumulated into (LayerNorm slices)
        self.zero = []                   # (start, length) inside the flat allocation
        for b in self.order:
            st, n = self.bucket[b]
            if not (layer_overwrite and b.startswith("layer")):
                self.zero.append((st, n))
            elif n > sizes[b]:
                self.zero.append((st + sizes[b], n - sizes[b]))
        self.numels = tuple(p.numel() for p in params)
        self.param = []                  # per parameter, in `names` order: (name, bucket, start, numel, shape)
        off = {b: 0 for b in sizes}
        for n, p in zip(names, params):
            b = bucket_of(n)
            st = self.bucket[b][0] + off[b]
            self.param.append((n, b, st, p.numel(), tuple(p.shape)))
            if layer_overwrite and b.startswith("layer"):
                if ".norm" in n:
                    self.zero.append((st, p.numel()))
                pad = (p.numel() + 3) // 4 * 4 - p.numel()
                if pad:                                   # alignment padding is part of the all-reduced buffer
                    self.zero.append((st + p.numel(), pad))
            off[b] += (p.numel() + 3) // 4 * 4
        # the flat allocation as consecutive pieces in memory order - per bucket its parameters' slots (padded to 4 elements) and
        # then its alignment tail - so that ONE split call yields every piece; piece_of[i] = the piece of parameter i
        by_bucket = {b: [] for b in self.order}
        for i, pr in enumerate(self.param):
            by_bucket[pr[1]].append(i)
        self.piece_sizes, self.piece_of = [], [0] * len(self.param)
        for b in self.order:
            used = 0
            for i in by_bucket[b]:
                slot = (self.param[i][3] + 3) // 4 * 4
                self.piece_of[i] = len(self.piece_sizes)
                self.piece_sizes.append(slot)
                used += slot
            if self.bucket[b][1] > used:
                self.piece_sizes.append(self.bucket[b][1] - used)
        # the zero fills in terms of those pieces (no fresh slices per pass): whole buckets, whole pieces, or a slot's padding
        starts, pos_ = [], 0
        for sz in self.piece_sizes:
            starts.append(pos_)
            pos_ += sz
        piece_at = {st: k for k, st in enumerate(starts)}
        bucket_at = {st: b for b, (st, n) in self.bucket.items()}
        self.zero_spec = []
        for st, n in self.zero:
            if st in bucket_at and self.bucket[bucket_at[st]][1] == n:
                self.zero_spec.append(("flat", bucket_at[st], 0))
            elif st in piece_at and self.piece_sizes[piece_at[st]] == n:
                self.zero_spec.append(("piece", piece_at[st], 0))
            else:   # the padding behind a parameter inside its slot, or a LayerNorm slice narrower than its slot
                k = max(kk for kk, s0 in enumerate(starts) if s0 <= st)
                self.zero_spec.append(("sub", k, (st - starts[k], st - starts[k] + n)))


class _GradBuckets:
    def __init__(self, rt, names, params, dev, bucket_of, layer_overwrite=False, extra_zero=(), layout=None):
        """layer_overwrite: the encoder layers' Linear weight/bias gradients are written by the kernels
        (TIMHIP_DESC_WGRAD_OVERWRITE); their buckets are left uninitialised except the LayerNorm slices,
        which the kernels accumulate into."""
        self.rt = rt
        lay = layout if layout is not None else _BucketLayout(names, params, bucket_of, layer_overwrite)
        self.order = lay.order
        # ONE allocation; parameter `.grad`s are views into it
        self.base = torch.empty(lay.total, dtype=torch.float32, device=dev)
        self.flat = {b: self.base[st:st + n] for b, (st, n) in lay.bucket.items()}
        self.views = {}
        self.members = {}   # bucket -> [(parameter, its gradient view)]: what a data-parallel hook needs to fold in earlier,
        #                     locally accumulated .grad values (tim_amd/dp.py: gradient accumulation under no_sync)
        base = self.base
        pieces = base.split(lay.piece_sizes)      # one call for all slots (a slice + a view per parameter cost 0.25 ms per pass)
        want_members = rt.bucket_hook is not None     # (only a data-parallel hook reads them)
        views, members = self.views, self.members
        for (n, b, st, numel, shape), p, k in zip(lay.param, params, lay.piece_of):
            pc = pieces[k]
            v = pc if pc.numel() == numel else pc[:numel]
            if len(shape) != 1:                      # (biases / LayerNorm vectors: the piece is the view)
                v = v.view(shape)
            views[n] = v
            if want_members:
                members.setdefault(b, []).append((p, v))
        # ONE multi-tensor launch for all the pieces that must start at zero
        # (`extra_zero`: other small buffers of the same backward pass - they ride in the same launch)
        accumulated = list(extra_zero)
        for kind, k, rng in lay.zero_spec:
            accumulated.append(self.flat[k] if kind == "flat" else (pieces[k] if kind == "piece" else pieces[k][rng[0]:rng[1]]))
        if accumulated:
            torch._foreach_zero_(accumulated)

    def done(self, bucket, ready=None):
        """`ready`: event after which the side-stream part of the bucket is complete (None: current stream)"""
        if self.rt.bucket_hook is not None and bucket in self.flat:
            self.rt.bucket_hook(bucket, self.flat[bucket], ready, self.members.get(bucket))

    def range_of(self, first, last):
        """the contiguous fp32 range covering buckets first..last (in completion order)"""
        a, b = self.flat[first], self.flat[last]
        start = (a.data_ptr() - self.base.data_ptr()) // 4
        end = (b.data_ptr() - self.base.data_ptr()) // 4 + b.numel()
        return self.base[start:end]


class TIM(nn.Module):
    _LAYER_GRAD_NAMES = ["self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight",
                         "self_attn.out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
                         "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"]

    def __init__(self,
                 num_class,
                 visual_input_dim=1024,
                 audio_input_dim=2304,
                 feat_drop=0.5,
                 seq_drop=0.5,
                 d_model=512,
                 feedforward_scale=4,
                 nhead=8,
                 num_layers=6,
                 enc_dropout=0.1,
                 input_modality="audio_visual",
                 data_modality="audio_visual",
                 num_feats=50,
                 include_verb_noun=True,
                 pool_features=False,
                 precision="fp16",
                 _variant="recognition"):
        super().__init__()
        if d_model % 32 != 0:
            raise ValueError("d_model must be a multiple of 32 for the gfx950 kernels")
        self.cfg = TimConfig(num_class=num_class, visual_input_dim=visual_input_dim, audio_input_dim=audio_input_dim,
                             feat_drop=feat_drop, seq_drop=seq_drop, d_model=d_model,
                             feedforward_scale=feedforward_scale, nhead=nhead, num_layers=num_layers,
                             enc_dropout=enc_dropout, input_modality=input_modality, data_modality=data_modality,
                             num_feats=num_feats, include_verb_noun=include_verb_noun, variant=_variant)
        cfg = self.cfg
        # attributes the reference exposes (tim.py:37-53)
        self.input_modality, self.data_modality = input_modality, data_modality
        self.visual_input_dim, self.audio_input_dim = visual_input_dim, audio_input_dim
        self.feat_drop, self.seq_drop = feat_drop, seq_drop
        self.d_model, self.dim_feedforward = d_model, d_model * feedforward_scale
        self.nhead, self.num_layers, self.enc_dropout = nhead, num_layers, enc_dropout
        self.num_feats = cfg.F  # the reference doubles num_feats for audio_visual (tim.py:88)
        self.num_class, self.include_verb_noun, self.pool_features = num_class, include_verb_noun, pool_features

        d, E = d_model, cfg.E
        self.time_mlp = nn.Sequential(nn.Linear(2, d), nn.ReLU(), nn.Linear(d, d), nn.ReLU(), nn.Linear(d, d),
                                      nn.ReLU(), nn.LayerNorm(d))
        self.feature_encoding = _FeatureEncodingParams(cfg)
        self.cls_head = _ClsHeadParams(cfg)
        if _variant == "detection":
            self.reg_head = _RegHeadParams(cfg)
        stack = _EncoderStackParams(_EncoderLayerParams(E, nhead, cfg.FF, enc_dropout), num_layers)
        if _variant == "detection":
            self.backbone = stack
        else:
            self.transformer_encoder = stack
        self._stack_prefix = "backbone" if _variant == "detection" else "transformer_encoder"
        self.drloc_mlp = nn.Sequential(nn.Linear(4 * d, d), nn.ReLU(), nn.Linear(d, d), nn.ReLU(), nn.Linear(d, 1))
        # AVGA pooling of the AVE recipe: an input pre-step in stock torch, outside the hot path (tim.py:137-144,155-156)
        self.pool = _AVGAParams(audio_input_dim, visual_input_dim, visual_input_dim) if pool_features else None

        self.rt = Runtime(precision)
        self._plans = {}
        self._ws = {}
        self._ws_pinned = False   # set by GraphedStep: workspaces captured in a graph are never freed
        self._ws_retired = []
        self._check_kernel_limits()
        self._encoder_param_names = [n for n, _ in self.named_parameters()
                                     if not (n.startswith("time_mlp.") or n.startswith("drloc_mlp.") or n.startswith("pool."))]

    def _check_kernel_limits(self):
        """Shapes the gfx950 kernels do not cover fail HERE, with the reason, not at the first forward."""
        cfg = self.cfg
        if cfg.F > 192:
            raise ValueError("num_feats gives %d feature tokens per window; the attention kernels keep a head's keys and values "
                             "in LDS and cover at most 192 (num_feats <= 96 for audio_visual, <= 192 for one modality)" % cfg.F)
        if cfg.E % cfg.nhead != 0:
            raise ValueError("nhead must divide the transformer width 2 * d_model = %d" % cfg.E)
        if self.rt.h16 and (cfg.E // cfg.nhead) not in (32, 64, 128):
            import warnings
            warnings.warn("head dimension %d: the MFMA attention kernels cover 32 / 64 / 128; this model runs the (much slower) "
                          "fp32-arithmetic attention kernels" % (cfg.E // cfg.nhead))

    # ---- dropout RNG state: NOT part of state_dict() (its keys must stay the reference's); a checkpoint that
    # should resume the mask sequence stores this beside it, as the reference's loop would store torch's RNG state
    def dropout_rng_state(self):
        return {"dropout_seed": int(self.rt.seed), "dropout_step": int(self.rt.step)}

    def set_dropout_rng_state(self, state):
        self.rt.seed = int(state["dropout_seed"])
        self.rt.step = int(state["dropout_step"])

    # ---- helpers used by tim_amd.functional --------------------------------------------------------
    def _encoder_param_list(self):
        """the parameters the encoder Function takes, in `_encoder_param_names` order.  Walking the module tree costs the host
        ~0.3 ms per call (a tenth of an eager step's issue time), so the list is kept; `nn.Module._apply` (.to / .cuda / .half
        keep the Parameter objects when torch's default `__future__` flags are in force, but may not) drops it, and so does
        `invalidate_param_cache()` - call that after REPLACING a parameter object inside the model (in-place updates,
        `load_state_dict`, optimizers need nothing)."""
        lst = self.__dict__.get("_enc_params")
        if lst is None:
            sd = dict(self.named_parameters())
            lst = [sd[n] for n in self._encoder_param_names]
            self.__dict__["_enc_params"] = lst
        return lst

    def invalidate_param_cache(self):
        self.__dict__.pop("_enc_params", None)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_enc_params", None)
        return super()._apply(fn, *args, **kwargs)

    def _plan(self, T, nv, na):
        key = (T, nv, na)
        p = self._plans.get(key)
        if p is None:
            p = EncoderPlan(self.cfg, T, nv, na)
            self._plans[key] = p
        return p

    def _workspace(self, nbytes, dev, slot="main"):
        """Scratch buffer of at least `nbytes`, cached per (device, slot) and grown on demand.  A captured HIP graph
        (tim_amd.graph.GraphedStep) holds raw pointers into the buffers it was captured with: once a graph exists on this
        model (`_ws_pinned`), an outgrown buffer is retired - kept alive, never handed back to the allocator - instead of
        freed, so a larger eager call between replays (validation at another batch size, a longer sequence) cannot make
        the graph write into memory that now belongs to other tensors."""
        key = (dev, slot)
        w = self._ws.get(key)
        if w is None or w.numel() < nbytes:
            if w is not None and self._ws_pinned:
                self._ws_retired.append(w)
            w = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
            self._ws[key] = w
        return w

    @staticmethod
    def _layer_params(rt, P, pre):
        from .functional import _f32c
        lsplit = rt.layer_split_for(P[pre + "linear1.weight"].shape[1], P[pre + "linear1.weight"].shape[0])

        def fwd_w(key, name):   # the forward operand of a Linear: plain 16-bit copy, or its split copy (Runtime.layer_split)
            return rt.weight_split(P[pre + name], mode=0) if key in lsplit else rt.weight(P[pre + name])
        keep = [fwd_w("in", "self_attn.in_proj_weight"), rt.weight(P[pre + "self_attn.in_proj_weight"], True),
                fwd_w("out", "self_attn.out_proj.weight"), rt.weight(P[pre + "self_attn.out_proj.weight"], True),
                fwd_w("l1", "linear1.weight"), rt.weight(P[pre + "linear1.weight"], True),
                fwd_w("l2", "linear2.weight"), rt.weight(P[pre + "linear2.weight"], True)]
        keep += [_f32c(P[pre + n]) for n in ("self_attn.in_proj_bias", "self_attn.out_proj.bias", "linear1.bias",
                                             "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight",
                                             "norm2.bias")]
        return L.TimLayerParams(*[t.data_ptr() for t in keep]), keep

    def invalidate_weights(self):
        """call after writing parameters through `.data` (see Runtime.invalidate_weights)"""
        self.rt.invalidate_weights()

    def _bucket_of(self, name):
        pre = self._stack_prefix + ".layers."
        if name.startswith(pre):
            return "layer" + name[len(pre):].split(".")[0]
        if name.startswith("feature_encoding."):
            return "front"
        return "heads"

    def _alloc_grad_buckets(self, names, params, dev, layer_overwrite=False, extra_zero=()):
        key = (bool(layer_overwrite), id(names), len(names))     # (`names`: the model's persistent parameter-name list)
        lay = self.__dict__.setdefault("_bucket_layouts", {}).get(key)
        # revalidated by the names at both ends (an id reused by another list) AND by every parameter's element count: a head
        # resized or replaced in place under the same name must not meet a stale layout - the kernels write through raw
        # pointers into these views (round-5 advisor finding).  ~100 numel() calls: 10 us per backward pass.
        numels = tuple(p.numel() for p in params)
        if lay is None or lay.param[0][0] != names[0] or lay.param[-1][0] != names[-1] or lay.numels != numels:
            lay = self._bucket_layouts[key] = _BucketLayout(names, params, self._bucket_of, layer_overwrite)
        return _GradBuckets(self.rt, names, params, dev, self._bucket_of, layer_overwrite, extra_zero, layout=lay)

    # ---- the reference's public interface ------------------------------------------------------------
    def forward_encoder(self, inputs, time_encodings, num_v_queries, num_a_queries):
        if self.pool is not None:
            inputs = [self.pool(inputs[1], inputs[0]), inputs[1]]
        outs = EncoderFn.apply(self, int(num_v_queries or 0), int(num_a_queries or 0), inputs[0], inputs[1],
                               time_encodings, *self._encoder_param_list())
        o = dict(zip(OUT_SLOTS, outs))
        return (o["verb"], o["noun"], o["action"], o["audio"]), o["feats"]

    def _time_mlp(self, times):
        m = self.time_mlp
        return TimeMlpFn.apply(self.rt, times, m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight,
                               m[4].bias, m[6].weight, m[6].bias)

    def forward(self, inputs, forward_type, time_encodings=None, num_v_queries=None, num_a_queries=None):
        if forward_type == "time_mlp":
            return self._time_mlp(inputs)
        elif forward_type == "encoder":
            return self.forward_encoder(inputs, time_encodings, num_v_queries, num_a_queries)
        elif forward_type == "drloc_mlp":
            # DRLoc auxiliary MLP (tim.py:129-135,190-191) on the GEMM kernels
            from .losses import drloc_mlp_forward
            return drloc_mlp_forward(self, inputs)
