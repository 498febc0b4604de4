"""Detection variant of TIM (detection/time_interval_machine/models/tim.py:17-430).

The encoder, heads and regression heads run on the HIP path; the multi-scale query
pyramid (tim.py:144-155) and the IoU labelling of queries (tim.py:157-270) are
no-grad fp32/int64 bookkeeping on a few hundred intervals and stay host-side torch
ops (adjacent, not a kernel target).
"""
import torch
import torch.nn.functional as F

from .functional import EncoderFn, OUT_SLOTS
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
                 precision="bf16"):
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

    # ---- tim.py:157-184
    def assign_positive_labels(self, modality, query_labels):
        ls = self.label_smoothing

        def smooth(lbl, n):
            return ((F.one_hot(lbl, n + 1) * ls) + ((1 - ls) / (n + 1)))[:, :-1]

        if modality == "visual":
            verb_labels = torch.empty(size=(0,)).to(device=query_labels.device)
            noun_labels = torch.empty(size=(0,)).to(device=query_labels.device)
            num_actions = self.num_class[0]
            if self.include_verb_noun:
                num_verbs, num_nouns, num_actions = self.num_class[0]
                query_labels[:, 0].masked_fill_(query_labels[:, 0] == -1, num_verbs)
                query_labels[:, 1].masked_fill_(query_labels[:, 1] == -1, num_nouns)
                verb_labels = smooth(query_labels[:, 0], num_verbs)
                noun_labels = smooth(query_labels[:, 1], num_nouns)
            query_labels[:, 2].masked_fill_(query_labels[:, 2] == -1, num_actions)
            return [verb_labels, noun_labels, smooth(query_labels[:, 2], num_actions)]
        num_actions = self.num_class[1]
        query_labels.masked_fill_(query_labels == -1, num_actions)
        return smooth(query_labels[:, -1], num_actions)

    # ---- tim.py:186-212
    def get_query_ious(self, queries, target_segs):
        q_s, q_e = queries[:, :, :, 0], queries[:, :, :, 1]
        g_s, g_e = target_segs[:, :, :, 0], target_segs[:, :, :, 1]
        neg = torch.abs(torch.clamp(g_s.min(dim=-1)[0], max=0.0))[:, :, None]
        q_s, q_e, g_s, g_e = q_s + neg, q_e + neg, g_s + neg, g_e + neg
        inter = torch.clamp(torch.minimum(q_e, g_e) - torch.maximum(q_s, g_s), min=0.0)
        unions = (g_e - g_s) + (q_e - q_s) - inter
        return inter / unions

    # ---- tim.py:214-270
    def label_queries(self, queries, target, modality, iou_threshold):
        if modality == "visual":
            target_segs = target['v_gt_segments']
            gt_labels = torch.stack([target['verb'], target['noun'], target['action']], dim=-1)
        else:
            target_segs = target['a_gt_segments']
            gt_labels = target['class_id'].unsqueeze(-1)
        nq, ng = queries.shape[1], target_segs.shape[1]
        q = queries[:, :, None].expand(-1, -1, ng, -1)
        t = target_segs[:, None].expand(-1, nq, -1, -1)
        lab = gt_labels[:, None].expand(-1, nq, -1, -1)
        ious = self.get_query_ious(q, t)
        idx = ious.argmax(-1)
        ious = torch.gather(ious, 2, idx[..., None]).squeeze(-1)
        query_targets = torch.gather(t, 2, idx[..., None, None].expand(-1, -1, 1, 2)).squeeze(2).clone()
        query_labels = torch.gather(lab, 2, idx[..., None, None].expand(-1, -1, 1, lab.shape[-1])).squeeze(2).clone()
        negatives = ious < iou_threshold
        query_targets.masked_fill_(negatives[:, :, None], float("inf"))
        query_labels.masked_fill_(negatives[:, :, None], -1)
        query_targets = torch.flatten(query_targets, 0, 1)
        query_labels = torch.flatten(query_labels, 0, 1)
        return query_targets, self.assign_positive_labels(modality, query_labels), torch.flatten(ious)

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
