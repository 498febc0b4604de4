"""Generate golden vectors by importing the reference (build container only).

    python tests/golden/make_golden.py recognition
    python tests/golden/make_golden.py detection

Two separate processes are needed because both reference variants use the
package name `time_interval_machine`.  Nothing of the
reference is written to the repo: the fixtures are plain numbers (outputs and
gradients of the reference module for inputs/weights that `tim_amd.synth`
regenerates from a seed).
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "recognition"

# ---- logging-only stand-ins so that models/tim.py imports ----
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
from tim_amd.config import TimConfig, named_config  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


def build_ref(cfg, dtype):
    kw = dict(visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim,
              feat_drop=cfg.feat_drop, seq_drop=cfg.seq_drop, d_model=cfg.d_model,
              nhead=cfg.nhead, num_layers=cfg.num_layers, enc_dropout=cfg.enc_dropout,
              input_modality=cfg.input_modality, data_modality=cfg.data_modality,
              num_feats=cfg.num_feats, include_verb_noun=cfg.include_verb_noun)
    if VARIANT == "detection":
        kw["feedfoward_scale"] = cfg.feedforward_scale
    else:
        kw["feedforward_scale"] = cfg.feedforward_scale
    m = TIM(cfg.num_class, **kw)
    return m.to(dtype).eval()


def load_synth(m, cfg, seed, dtype):
    sd = synth.make_state_dict(cfg, seed=seed, dtype=np.float64)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(k for k in ref_sd.keys()), "order"
    assert set(ref_sd.keys()) == set(sd.keys()), (set(ref_sd) ^ set(sd))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == sd[k].shape, (k, v.shape, sd[k].shape)
    m.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in sd.items()})
    return sd


def run_rec(cfg, B, nv, na, seed, dtype, with_grads, layer_outs=False):
    m = build_ref(cfg, dtype)
    load_synth(m, cfg, seed, dtype)
    inp = synth.make_inputs(cfg, B, nv, na, seed=seed, dtype=np.float64)
    vis = torch.from_numpy(inp["visual"]).to(dtype).requires_grad_(inp["visual"].ndim == 3)
    aud = torch.from_numpy(inp["audio"]).to(dtype).requires_grad_(inp["audio"].ndim == 3)
    times = torch.from_numpy(inp["times"]).to(dtype).requires_grad_(True)
    outs = {}
    hooks = []
    if layer_outs:
        for i, lyr in enumerate(m.transformer_encoder.layers):
            hooks.append(lyr.register_forward_hook(
                lambda mod, a, o, i=i: outs.__setitem__("layer%d" % i, o[0].detach().transpose(0, 1))))
        hooks.append(m.feature_encoding.register_forward_hook(
            lambda mod, a, o: outs.__setitem__("seq", o.detach().transpose(0, 1))))
    te = m(times, "time_mlp")
    (verb, noun, action, audio), feats = m([vis, aud], "encoder", te, nv, na)
    for h in hooks:
        h.remove()
    outs.update(te=te, feats=feats)
    for k, v in (("verb", verb), ("noun", noun), ("action", action), ("audio", audio)):
        if v is not None:
            outs[k] = v
    res = {"out/" + k: v.detach().numpy() for k, v in outs.items()}
    if with_grads:
        heads = {k: outs[k] for k in ("verb", "noun", "action", "audio", "feats") if k in outs}
        R = synth.make_cotangents(cfg, B, nv, na, {k: tuple(v.shape) for k, v in heads.items()},
                                  seed=seed, dtype=np.float64)
        loss = sum((heads[k] * torch.from_numpy(R[k]).to(dtype)).sum() for k in heads)
        loss.backward()
        res["loss"] = np.array(loss.item())
        for k, p in m.named_parameters():
            if p.grad is not None:
                res["grad/" + k] = p.grad.numpy()
        for k, t in (("visual", vis), ("audio", aud), ("times", times)):
            if t.grad is not None:
                res["gin/" + k] = t.grad.numpy()
    return res


def run_det(cfg, B, seed, dtype, with_grads):
    m = build_ref(cfg, dtype)
    m.train_pool = m.train_pool.to(dtype)
    m.inference_queries = m.inference_queries.to(dtype)
    load_synth(m, cfg, seed, dtype)
    nq = m.num_queries
    inp = synth.make_inputs(cfg, B, 0, 0, seed=seed, dtype=np.float64)
    vis = torch.from_numpy(inp["visual"]).to(dtype).requires_grad_(inp["visual"].ndim == 3)
    aud = torch.from_numpy(inp["audio"]).to(dtype).requires_grad_(inp["audio"].ndim == 3)
    times = torch.from_numpy(inp["times"]).to(dtype)
    (cls, reg, feats), _, _, (vq, aq), _ = m([vis, aud], "encoder", times, None, label_queries=False)
    outs = {"feats": feats}
    for k, v in zip(("verb", "noun", "action", "audio"), cls):
        if v is not None:
            outs[k] = v
    for k, v in zip(("reg_visual", "reg_audio"), reg):
        if v is not None:
            outs[k] = v
    res = {"out/" + k: v.detach().numpy() for k, v in outs.items()}
    res["queries"] = m.inference_queries.numpy()
    res["train_pool"] = m.train_pool.numpy()
    if with_grads:
        R = synth.make_cotangents(cfg, B, 0, 0, {k: tuple(v.shape) for k, v in outs.items()},
                                  seed=seed, dtype=np.float64)
        loss = sum((outs[k] * torch.from_numpy(R[k]).to(dtype)).sum() for k in outs)
        loss.backward()
        res["loss"] = np.array(loss.item())
        for k, p in m.named_parameters():
            if p.grad is not None:
                res["grad/" + k] = p.grad.numpy()
        for k, t in (("visual", vis), ("audio", aud)):
            if t.grad is not None:
                res["gin/" + k] = t.grad.numpy()
    return res, nq


def summarize(res, nslice=8):
    """Large configs: keep logit slices + per-tensor statistics only."""
    out = {}
    for k, v in res.items():
        v = np.asarray(v, dtype=np.float64)
        if k.startswith("out/") and v.ndim == 2 and k != "out/te":
            out[k + "/slice"] = v[:, :nslice].astype(np.float32)
        out[k + "/stats"] = np.array([v.sum(), np.abs(v).sum(), np.abs(v).max(),
                                      np.sqrt((v * v).sum())])
    return out


def tiny(input_modality, data_modality, vn, num_class=None):
    c = named_config("tiny")
    c.input_modality, c.data_modality, c.include_verb_noun = input_modality, data_modality, vn
    c.variant = VARIANT
    if num_class is not None:
        c.num_class = num_class
    elif not vn:
        c.num_class = [13, 5]
    return c


def main():
    keys = {}
    if VARIANT == "recognition":
        # (i) state_dict key/shape lists for every modality combo
        for im in ("audio_visual", "visual", "audio"):
            for dm in ("audio_visual", "visual", "audio"):
                for vn in (True, False):
                    if im != "audio_visual" and dm != im:
                        continue  # single-modality models only make sense on their own data
                    cfg = tiny(im, dm, vn)
                    m = build_ref(cfg, torch.float32)
                    keys["%s/%s/%d" % (im, dm, int(vn))] = [
                        [k, list(v.shape)] for k, v in m.state_dict().items()]
        # parameter count of the EPIC A+V model (SURVEY: 58,303,640)
        m = build_ref(named_config("C2a"), torch.float32)
        keys["_count_C2a"] = sum(p.numel() for p in m.parameters())
        json.dump(keys, open(os.path.join(HERE, "keys_recognition.json"), "w"), indent=0)

        # (ii) tiny goldens: full tensors, fp64, with gradients
        combos = [("audio_visual", "audio_visual", True, 4, 2, True),
                  ("audio_visual", "audio_visual", False, 4, 2, False),
                  ("audio_visual", "visual", True, 3, 0, False),
                  ("audio_visual", "audio", True, 0, 3, False),
                  ("audio_visual", "audio_visual", True, 4, 0, False),  # Na=0 edge (head.py:18)
                  ("visual", "visual", True, 5, 0, True),
                  ("visual", "visual", False, 5, 0, False),
                  ("audio", "audio", True, 0, 3, True)]
        for im, dm, vn, nv, na, full in combos:
            cfg = tiny(im, dm, vn)
            res = run_rec(cfg, 3, nv, na, seed=1, dtype=torch.float64, with_grads=True,
                          layer_outs=True)
            if not full:  # keep gradients as norms only
                res = {k: (v if not k.startswith("grad/") else np.array(np.sqrt((v * v).sum())))
                       for k, v in res.items()}
            name = "tiny_rec_%s_%s_vn%d_nv%d_na%d.npz" % (im, dm, int(vn), nv, na)
            np.savez_compressed(os.path.join(HERE, name), **res)
            print(name, "loss", float(res["loss"]))
        # (iii) C1 / C2a / C3 at small batch, fp32 reference as the loops run it (eval, no autocast)
        for cname, B, nv, na in (("C1", 2, 10, 0), ("C2a", 2, 15, 10), ("C3", 2, 15, 10)):
            cfg = named_config(cname)
            res = run_rec(cfg, B, nv, na, seed=2, dtype=torch.float32, with_grads=True)
            np.savez_compressed(os.path.join(HERE, "%s_rec_summary.npz" % cname), **summarize(res))
            print(cname, "loss", float(res["loss"]))
    else:
        for im, dm, nc in (("audio_visual", "visual", (13, 5)),
                           ("audio_visual", "audio_visual", (13, 5)),
                           ("audio_visual", "audio_visual", [[7, 11, 13], 5]),
                           ("visual", "visual", [[7, 11, 13], 5]),
                           ("audio", "audio", (13, 5))):
            cfg = tiny(im, dm, isinstance(nc[0], list), num_class=nc)
            m = build_ref(cfg, torch.float32)
            tag = "%s/%s/%s" % (im, dm, "vn" if isinstance(nc[0], list) else "single")
            keys[tag] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
            res, nq = run_det(cfg, 2, seed=3, dtype=torch.float64, with_grads=True)
            res = {k: (v if not k.startswith("grad/") or (im, dm) == ("audio_visual", "visual")
                       else np.array(np.sqrt((v * v).sum()))) for k, v in res.items()}
            name = "tiny_det_%s_%s_%s.npz" % (im, dm, "vn" if isinstance(nc[0], list) else "single")
            np.savez_compressed(os.path.join(HERE, name), **res)
            print(name, "nq", nq, "loss", float(res["loss"]))
        json.dump(keys, open(os.path.join(HERE, "keys_detection.json"), "w"), indent=0)
        cfg = named_config("C4")
        res, nq = run_det(cfg, 1, seed=4, dtype=torch.float32, with_grads=False)
        res.pop("train_pool")
        res.pop("queries")
        np.savez_compressed(os.path.join(HERE, "C4_det_summary.npz"), **summarize(res))
        print("C4 nq", nq)


if __name__ == "__main__":
    main()
