"""Detection variant of TIM (detection/time_interval_machine/models/tim.py:17-430).

The encoder, heads and regression heads run on the HIP path, and so does the IoU labelling
of the queries (tim.py:157-270 -> csrc/labels.hip: fp32 interval arithmetic bit-identical to
the reference's).  The multi-scale query pyramid itself
(tim.py:144-155) is a constant built once on the host.
"""
import numpy as np
import torch

from ._lib import call, ptr
from .functional import EncoderFn, OUT_SLOTS, _require_gpu, _stream
from .tim import TIM as _TIMBase


class TIM(_TIMBase):
    def __init__(self,
                 num_class,
                 visual_input_dim=1024,
                 audio_input_dim=2304,
                 feat_drop=0.5,
                 seq_drop=0.5,
                 d_model=512,
                 feedfoward_scale=4,  # (sic) the detection reference spells it this way, tim.py:25
                 nhead=8,
                 num_layers=6,
                 enc_dropout=0.1,
                 input_modality="audio_visual",
                 data_modality="audio_visual",
                 num_feats=50,
                 include_verb_noun=True,
                 iou_threshold=0.25,
                 label_smoothing=0.9,
                 precision="fp16"):
        super().__init__(num_class, visual_input_dim, audio_input_dim, feat_drop, seq_drop, d_model,
                         feedfoward_scale, nhead, num_layers, enc_dropout, input_modality, data_modality,
                         num_feats, include_verb_noun, False, precision, _variant="detection")
        self.iou_threshold = iou_threshold
        self.label_smoothing = label_smoothing
        self.train_pool = self.generate_queries(query_size=0.005)
        self.inference_queries = self.generate_queries(query_size=0.01)
        self.num_queries = self.inference_queries.shape[1]

    # ---- tim.py:144-155
    def generate_queries(self, query_size):
        queries = []
        while query_size < 1.0:
            start_times = torch.arange(0.0, 1.0, step=query_size / 2)
            layer_times = torch.stack([start_times, start_times + query_size], dim=-1)
            queries.append(torch.round(layer_times, decimals=3))
            query_size *= 2
        return torch.concat(queries, dim=0).unsqueeze(0)

    # ---- tim.py:157-270: IoU matching and label-smoothed targets, on the device (csrc/labels.hip)
    def assign_positive_labels(self, modality, query_labels):
        """query_labels [R, NL] int64 with -1 for negatives -> the reference's structure: [verb, noun, action] smoothed target
        matrices for "visual" (the first two empty when the model has no verb / noun heads), one matrix for "audio"."""
        _require_gpu(query_labels, "assign_positive_labels")
        ql = query_labels.contiguous()
        R, NL = ql.shape
        ls = float(self.label_smoothing)

        def smooth(col, n):
            out = torch.empty((R, n), dtype=torch.float32, device=ql.device)
            if R == 0:
                return out
            base = np.float32((1.0 - ls) / (n + 1))            # the Python float the reference adds, rounded to fp32 by the add
            on = np.float32(np.float32(ls) + base)             # one_hot * smoothing is an fp32 tensor
            call("timhip_smooth_one_hot", ptr(ql), NL, col, R, n, float(on), float(base), ptr(out), _stream())
            return out

        if modality == "visual":
            verb_labels = torch.empty(size=(0,)).to(device=ql.device)
            noun_labels = torch.empty(size=(0,)).to(device=ql.device)
            num_actions = self.num_class[0]
            if self.include_verb_noun:
                num_verbs, num_nouns, num_actions = self.num_class[0]
                verb_labels, noun_labels = smooth(0, num_verbs), smooth(1, num_nouns)
            return [verb_labels, noun_labels, smooth(2, num_actions)]
        return smooth(NL - 1, self.num_class[1])

    def label_queries(self, queries, target, modality, iou_threshold):
        """queries [B, Nq, 2]; returns (query_targets [B*Nq, 2], query_labels, query_ious [B*Nq]) as tim.py:214-270 does:
        the matched ground-truth segment (+inf for queries under the IoU threshold), the smoothed classification targets and
        the IoU of the match."""
        if modality == "visual":
            target_segs = target['v_gt_segments']
            gt_labels = torch.stack([target['verb'], target['noun'], target['action']], dim=-1)
        else:
            target_segs = target['a_gt_segments']
            gt_labels = target['class_id'].unsqueeze(-1)
        _require_gpu(queries, "label_queries")
        dev = queries.device
        B, Nq = queries.shape[:2]
        Ng, NL = target_segs.shape[1], gt_labels.shape[-1]
        if Ng == 0:
            raise ValueError("label_queries: the batch holds no (padded) ground-truth segment slots")
        q = queries.detach().to(torch.float32).contiguous()
        sg = target_segs.detach().to(device=dev, dtype=torch.float32).contiguous()
        lab = gt_labels.detach().to(device=dev, dtype=torch.int64).contiguous()
        query_targets = torch.empty((B * Nq, 2), dtype=torch.float32, device=dev)
        query_ious = torch.empty((B * Nq,), dtype=torch.float32, device=dev)
        query_labels = torch.empty((B * Nq, NL), dtype=torch.int64, device=dev)
        call("timhip_label_queries", ptr(q), ptr(sg), ptr(lab), B, Nq, Ng, NL, float(np.float32(iou_threshold)),
             ptr(query_targets), ptr(query_ious), ptr(query_labels), _stream())
        return query_targets, self.assign_positive_labels(modality, query_labels), query_ious

    # ---- tim.py:272-400
    def _run(self, inputs, feature_times, target, train, label_queries):
        v_offsets = a_offsets = torch.empty(0, 2)
        v_labels = a_labels = torch.empty(0, 4)
        num_v = num_a = 0
        v_queries = a_queries = v_ious = a_ious = None
        all_times = feature_times
        dev = feature_times.device
        Bsz = feature_times.shape[0]

        def on_dev(name):
            # the query pyramids are plain CPU tensors in the reference (moved on every call, det tim.py:296-300); here each is
            # copied to the device once - no per-step host-to-device copy, and the forward stays capturable in a HIP graph
            cache = self.__dict__.setdefault("_pyramid_dev", {})
            t = cache.get((name, dev))
            if t is None:
                t = getattr(self, name).to(device=dev)
                cache[(name, dev)] = t
            return t

        def draw():
            if train:
                if dev.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    sel = torch.randperm(self.train_pool.shape[1], device=dev)[:self.num_queries]   # graph-safe generator
                else:
                    sel = torch.randperm(self.train_pool.shape[1])[:self.num_queries].to(dev)     # the reference's CPU draw
                return on_dev("train_pool")[:, sel.long()].repeat(Bsz, 1, 1)
            return on_dev("inference_queries").repeat(Bsz, 1, 1)

        if "visual" in self.data_modality:
            v_queries = draw()
            num_v = v_queries.shape[1]
            if train or label_queries:
                v_offsets, v_labels, v_ious = self.label_queries(v_queries, target, "visual", self.iou_threshold)
            all_times = torch.concat([all_times, v_queries.to(all_times.dtype)], dim=1)
            v_queries = torch.flatten(v_queries, 0, 1)
        if "audio" in self.data_modality:
            a_queries = draw()
            num_a = a_queries.shape[1]
            if train or label_queries:
                a_offsets, a_labels, a_ious = self.label_queries(a_queries, target, "audio", self.iou_threshold)
            all_times = torch.concat([all_times, a_queries.to(all_times.dtype)], dim=1)
            a_queries = torch.flatten(a_queries, 0, 1)

        time_encodings = self._time_mlp(all_times)
        outs = EncoderFn.apply(self, num_v, num_a, inputs[0], inputs[1], time_encodings, *self._encoder_param_list())
        o = dict(zip(OUT_SLOTS, outs))
        cls_scores = (o["verb"], o["noun"], o["action"], o["audio"])
        reg_scores = (o["reg_visual"], o["reg_audio"])
        return (cls_scores, reg_scores, o["feats"]), (v_offsets, a_offsets), (v_labels, a_labels), \
            (v_queries, a_queries), (v_ious, a_ious)

    def forward_train(self, inputs, feature_times, target):
        return self._run(inputs, feature_times, target, True, True)

    def forward_inference(self, inputs, feature_times, target, label_queries=False):
        return self._run(inputs, feature_times, target, False, label_queries)

    def forward_encoder(self, inputs, feature_times, target, label_queries=False):
        if self.training:
            return self.forward_train(inputs, feature_times, target)
        return self.forward_inference(inputs, feature_times, target, label_queries)

    def forward(self, inputs, forward_type, feature_times=None, target=None, label_queries=False):
        if forward_type == "encoder":
            return self.forward_encoder(inputs, feature_times, target, label_queries)
        elif forward_type == "drloc_mlp":
            from .losses import drloc_mlp_forward
            return drloc_mlp_forward(self, inputs)
