"""Training-mode golden vectors from the imported reference (build container only).

    python tests/golden/make_golden_train.py recognition
    python tests/golden/make_golden_train.py detection

The reference draws its dropout masks from torch's global generator inside `torch.nn.functional.dropout`
(`nn.Dropout.forward` and `F.multi_head_attention_forward` both resolve that name at call time).  Here the name is replaced
by a recorder: it draws a keep-mask from a seeded generator, stores it, and applies it with the 1/(1-p) scale - the same
arithmetic `F.dropout(training=True)` states, with a mask that can be written down.  The reference module then runs in
`.train()` in fp64; the fixture holds the masks in call order (rec encodings.py:140-153,249; transformers.py:73,102-109),
every output and every gradient.  `tests/test_oracle_golden.py::test_tiny_train_mode_fp64` feeds the masks to
`oracle.tim_oracle.forward(..., masks=...)`.

Nothing of the reference is written to the repo: the fixtures are numbers.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "recognition"

sj = types.ModuleType("simplejson")
sj.dumps = lambda *a, **k: ""
sys.modules["simplejson"] = sj
for name in ("fvcore", "fvcore.common", "fvcore.common.file_io"):
    sys.modules[name] = types.ModuleType(name)


class _PM:
    open = staticmethod(open)


sys.modules["fvcore.common.file_io"].PathManager = _PM
sys.path.insert(0, "/root/reference/" + VARIANT)
from time_interval_machine.models.tim import TIM  # noqa: E402

from tim_amd import synth  # noqa: E402
from tim_amd.config import named_config  # noqa: E402

torch.set_num_threads(8)


class DropoutRecorder:
    """stands in for torch.nn.functional.dropout while the reference runs"""

    def __init__(self, seed):
        self.gen = torch.Generator().manual_seed(seed)
        self.calls = []          # (p, keep-mask uint8) in call order
        self._orig = torch.nn.functional.dropout

    def __call__(self, input, p=0.5, training=True, inplace=False):
        assert training and not inplace
        keep = (torch.rand(input.shape, generator=self.gen, dtype=torch.float64) >= p)
        self.calls.append((float(p), keep.to(torch.uint8)))
        if p == 0.0:
            return input
        return input * keep.to(input.dtype) * (1.0 / (1.0 - p))

    def __enter__(self):
        torch.nn.functional.dropout = self
        return self

    def __exit__(self, *a):
        torch.nn.functional.dropout = self._orig


def build_ref(cfg):
    kw = dict(visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim,
              feat_drop=cfg.feat_drop, seq_drop=cfg.seq_drop, d_model=cfg.d_model,
              nhead=cfg.nhead, num_layers=cfg.num_layers, enc_dropout=cfg.enc_dropout,
              input_modality=cfg.input_modality, data_modality=cfg.data_modality,
              num_feats=cfg.num_feats, include_verb_noun=cfg.include_verb_noun)
    kw["feedfoward_scale" if VARIANT == "detection" else "feedforward_scale"] = cfg.feedforward_scale
    return TIM(cfg.num_class, **kw).double().train()


def load_synth(m, cfg, seed):
    sd = synth.make_state_dict(cfg, seed=seed, dtype=np.float64)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})


def name_masks(cfg, calls, B, S):
    """call order -> the oracle's mask keys.  Layer masks arrive as [S,B,.] (the reference runs sequence-first) and the
    attention mask as the dense [B*H,S,S] matrix; they are stored batch-first, the attention mask as [B,H,S,F+1] with the
    query rows' own column (the diagonal) in column F - the only entries the structured form can see: every other entry of
    the dense mask multiplies an exact zero of the masked softmax."""
    it = iter(calls)
    out = {}
    F, Hh = cfg.F, cfg.nhead

    def take(p_expect, shape):
        p, mk = next(it)
        assert abs(p - p_expect) < 1e-12 and tuple(mk.shape) == tuple(shape), (p, p_expect, mk.shape, shape)
        return mk

    if cfg.input_modality in ("audio_visual", "visual"):
        out["feat_visual"] = take(cfg.feat_drop, (B, cfg.num_feats, cfg.visual_input_dim))
    if cfg.input_modality in ("audio_visual", "audio"):
        out["feat_audio"] = take(cfg.feat_drop, (B, cfg.num_feats, cfg.audio_input_dim))
    out["seq"] = take(cfg.seq_drop, (B, S, cfg.E))
    for l in range(cfg.num_layers):
        a = take(cfg.enc_dropout, (B * Hh, S, S)).reshape(B, Hh, S, S)
        diag = torch.diagonal(a, dim1=2, dim2=3)                            # [B,H,S]
        out["l%d_attn" % l] = torch.cat([a[..., :F], diag[..., None]], -1)  # rows < F never read column F
        out["l%d_drop1" % l] = take(cfg.enc_dropout, (S, B, cfg.E)).transpose(0, 1).contiguous()
        out["l%d_ffn" % l] = take(cfg.enc_dropout, (S, B, cfg.FF)).transpose(0, 1).contiguous()
        out["l%d_drop2" % l] = take(cfg.enc_dropout, (S, B, cfg.E)).transpose(0, 1).contiguous()
    assert next(it, None) is None, "unclaimed dropout call"
    return out


def grads_of(m, outs, cfg, B, nv, na, seed, leaves):
    R = synth.make_cotangents(cfg, B, nv, na, {k: tuple(v.shape) for k, v in outs.items()}, seed=seed, dtype=np.float64)
    loss = sum((outs[k] * torch.from_numpy(R[k])).sum() for k in outs)
    loss.backward()
    res = {"loss": np.array(loss.item())}
    for k, p in m.named_parameters():
        if p.grad is not None:
            res["grad/" + k] = p.grad.numpy()
    for k, t in leaves.items():
        if t.grad is not None:
            res["gin/" + k] = t.grad.numpy()
    return res


def run_rec(cfg, B, nv, na, seed, mask_seed):
    m = build_ref(cfg)
    load_synth(m, cfg, seed)
    inp = synth.make_inputs(cfg, B, nv, na, seed=seed, dtype=np.float64)
    vis = torch.from_numpy(inp["visual"]).requires_grad_(inp["visual"].ndim == 3)
    aud = torch.from_numpy(inp["audio"]).requires_grad_(inp["audio"].ndim == 3)
    times = torch.from_numpy(inp["times"]).requires_grad_(True)
    with DropoutRecorder(mask_seed) as rec:
        te = m(times, "time_mlp")
        (verb, noun, action, audio), feats = m([vis, aud], "encoder", te, nv, na)
    outs = {"feats": feats}
    for k, v in (("verb", verb), ("noun", noun), ("action", action), ("audio", audio)):
        if v is not None:
            outs[k] = v
    S = cfg.F + cfg.num_queries(nv, na)
    res = {"mask/" + k: v.numpy() for k, v in name_masks(cfg, rec.calls, B, S).items()}
    res.update({"out/" + k: v.detach().numpy() for k, v in outs.items()})
    res["out/te"] = te.detach().numpy()
    res.update(grads_of(m, outs, cfg, B, nv, na, seed, {"visual": vis, "audio": aud, "times": times}))
    return res, len(rec.calls)


def run_det_train(cfg, B, seed, mask_seed):
    """`forward_train` (det tim.py:272-337): the queries are drawn inside the model (torch.randperm, :281) and returned;
    the fixture stores them, the labels the model computed for them, and the usual outputs / gradients."""
    m = build_ref(cfg)
    m.train_pool = m.train_pool.double()
    m.inference_queries = m.inference_queries.double()
    load_synth(m, cfg, seed)
    inp = synth.make_inputs(cfg, B, 0, 0, seed=seed, dtype=np.float64)
    vis = torch.from_numpy(inp["visual"]).requires_grad_(inp["visual"].ndim == 3)
    aud = torch.from_numpy(inp["audio"]).requires_grad_(inp["audio"].ndim == 3)
    times = torch.from_numpy(inp["times"])
    g = torch.Generator().manual_seed(seed)
    ngt = 3
    st = torch.rand(B, ngt, generator=g, dtype=torch.float64) * 0.8
    seg = torch.stack([st, st + 0.02 + 0.3 * torch.rand(B, ngt, generator=g, dtype=torch.float64)], -1)
    ncls = cfg.num_class
    lab = torch.stack([torch.randint(0, int(ncls[0]), (B, ngt), generator=g)] * 3
                      + [torch.randint(0, int(ncls[1]), (B, ngt), generator=g)], -1)
    target = {"v_gt_segments": seg, "a_gt_segments": seg.clone(), "verb": lab[..., 0], "noun": lab[..., 1],
              "action": lab[..., 2], "class_id": lab[..., 3]}
    torch.manual_seed(mask_seed)        # the model's own randperm
    with DropoutRecorder(mask_seed) as rec:
        (cls, reg, feats), offs, labs, (vq, aq), ious = m([vis, aud], "encoder", times, target, label_queries=True)
    nq = m.num_queries
    outs = {"feats": feats}
    for k, v in zip(("verb", "noun", "action", "audio"), cls):
        if v is not None:
            outs[k] = v
    for k, v in zip(("reg_visual", "reg_audio"), reg):
        if v is not None:
            outs[k] = v
    nv = nq if cfg.has_visual_queries else 0
    na = nq if cfg.has_audio_queries else 0
    S = cfg.F + nv + na
    res = {"mask/" + k: v.numpy() for k, v in name_masks(cfg, rec.calls, B, S).items()}
    res.update({"out/" + k: v.detach().numpy() for k, v in outs.items()})
    if vq is not None:
        res["v_queries"] = vq.reshape(B, nq, 2).numpy()
        res["v_offsets"] = offs[0].numpy()
        res["v_ious"] = ious[0].numpy()
    if aq is not None:
        res["a_queries"] = aq.reshape(B, nq, 2).numpy()
        res["a_offsets"] = offs[1].numpy()
        res["a_ious"] = ious[1].numpy()
    res["gt_segments"] = seg.numpy()
    res.update(grads_of(m, outs, cfg, B, 0, 0, seed, {"visual": vis, "audio": aud}))
    return res, len(rec.calls)


def tiny(im, dm, vn, num_class=None):
    c = named_config("tiny")
    c.input_modality, c.data_modality, c.include_verb_noun = im, dm, vn
    c.variant = VARIANT
    if num_class is not None:
        c.num_class = num_class
    elif not vn:
        c.num_class = [13, 5]
    return c


def pack(res):
    """masks as bit-packed uint8 (shape stored beside), everything else fp64"""
    out = {}
    for k, v in res.items():
        if k.startswith("mask/"):
            out[k] = np.packbits(v.reshape(-1))
            out["shape/" + k[5:]] = np.array(v.shape)
        else:
            out[k] = v
    return out


def main():
    if VARIANT == "recognition":
        combos = [("audio_visual", "audio_visual", True, 4, 2),
                  ("audio_visual", "audio_visual", False, 4, 2),
                  ("visual", "visual", True, 5, 0),
                  ("audio", "audio", True, 0, 3)]
        for i, (im, dm, vn, nv, na) in enumerate(combos):
            cfg = tiny(im, dm, vn)
            res, ncalls = run_rec(cfg, 3, nv, na, seed=5, mask_seed=100 + i)
            name = "tiny_train_rec_%s_%s_vn%d_nv%d_na%d.npz" % (im, dm, int(vn), nv, na)
            np.savez_compressed(os.path.join(HERE, name), **pack(res))
            print(name, "dropout calls", ncalls, "loss", float(res["loss"]))
    else:
        cfg = tiny("audio_visual", "visual", False, num_class=(13, 5))
        res, ncalls = run_det_train(cfg, 2, seed=6, mask_seed=200)
        name = "tiny_train_det_audio_visual_visual_single.npz"
        np.savez_compressed(os.path.join(HERE, name), **pack(res))
        print(name, "dropout calls", ncalls, "loss", float(res["loss"]), "positives",
              int(np.isfinite(res["v_offsets"][:, 0]).sum()))


if __name__ == "__main__":
    main()
