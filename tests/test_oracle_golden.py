"""The oracle (oracle/tim_oracle.py) against golden vectors produced by the
imported reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import tim_oracle as O
from tim_amd.config import named_config
from tests import helpers as H

TOL64 = 1e-11  # fp64 restatement vs fp64 reference


@pytest.mark.parametrize("fname,im,dm,vn,nv,na", H.rec_golden_cases())
def test_tiny_recognition_fp64(fname, im, dm, vn, nv, na):
    g = np.load(os.path.join(H.GOLDEN, fname))
    cfg = H.tiny_cfg("recognition", im, dm, vn)
    sd, inp = H.synth_torch(cfg, 3, nv, na, seed=1, dtype=torch.float64)
    for t in sd.values():
        t.requires_grad_(True)
    leaves = {k: inp[k].clone().requires_grad_(inp[k].ndim == 3) for k in ("visual", "audio", "times")}
    te = O.time_mlp(sd, leaves["times"])
    cls, feats, layers = O.encoder(sd, cfg, leaves["visual"], leaves["audio"], te, nv, na,
                                   return_layers=True)
    outs = H.named_outputs(cls, feats)
    np.testing.assert_allclose(te.detach().numpy(), g["out/te"], atol=TOL64, rtol=0)
    np.testing.assert_allclose(layers[0].detach().numpy(), g["out/seq"], atol=TOL64, rtol=0)
    for l in range(cfg.num_layers):
        np.testing.assert_allclose(layers[l + 1].detach().numpy(), g["out/layer%d" % l], atol=TOL64, rtol=0)
    for k, v in outs.items():
        assert v.shape == g["out/" + k].shape, k
        np.testing.assert_allclose(v.detach().numpy(), g["out/" + k], atol=TOL64, rtol=0, err_msg=k)
    assert set(outs) == {k[4:] for k in g.files if k.startswith("out/")} - {"te", "seq"} - {
        "layer%d" % l for l in range(cfg.num_layers)}
    R = H.cotangents(cfg, 3, nv, na, outs, seed=1, dtype=torch.float64)
    loss = sum((outs[k] * R[k]).sum() for k in outs)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-9
    for k in g.files:
        if k.startswith("grad/"):
            gr = sd[k[5:]].grad
            ref = g[k]
            if ref.ndim == 0:  # stored as a norm only
                assert abs(gr.norm().item() - float(ref)) <= 1e-9 * max(1.0, float(ref)), k
            else:
                np.testing.assert_allclose(gr.numpy(), ref, atol=1e-10, rtol=0, err_msg=k)
        if k.startswith("gin/"):
            np.testing.assert_allclose(leaves[k[4:]].grad.numpy(), g[k], atol=1e-10, rtol=0, err_msg=k)


@pytest.mark.parametrize("im,dm,nc,tag", H.DET_CASES)
def test_tiny_detection_fp64(im, dm, nc, tag):
    g = np.load(os.path.join(H.GOLDEN, "tiny_det_%s_%s_%s.npz" % (im, dm, tag)))
    cfg = H.tiny_cfg("detection", im, dm, tag == "vn", num_class=nc)
    sd, inp = H.synth_torch(cfg, 2, 0, 0, seed=3, dtype=torch.float64)
    for t in sd.values():
        t.requires_grad_(True)
    q = O.generate_queries(0.01).to(torch.float64)
    np.testing.assert_array_equal(q.numpy(), g["queries"])
    np.testing.assert_array_equal(O.generate_queries(0.005).double().numpy(), g["train_pool"])
    nq = q.shape[1]
    assert nq == 399  # det tim.py:140-142, SURVEY 3.4
    nv = nq if cfg.has_visual_queries else 0
    na = nq if cfg.has_audio_queries else 0
    times = inp["times"]
    if nv:
        times = torch.cat([times, q.expand(2, -1, -1)], 1)
    if na:
        times = torch.cat([times, q.expand(2, -1, -1)], 1)
    vis = inp["visual"].clone().requires_grad_(inp["visual"].ndim == 3)
    aud = inp["audio"].clone().requires_grad_(inp["audio"].ndim == 3)
    cls, feats, reg = O.forward(sd, cfg, vis, aud, times, nv, na)
    outs = H.named_outputs(cls, feats, reg)
    assert set(outs) == {k[4:] for k in g.files if k.startswith("out/")}
    for k, v in outs.items():
        np.testing.assert_allclose(v.detach().numpy(), g["out/" + k], atol=TOL64, rtol=0, err_msg=k)
    R = H.cotangents(cfg, 2, 0, 0, outs, seed=3, dtype=torch.float64)
    loss = sum((outs[k] * R[k]).sum() for k in outs)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-9
    for k in g.files:
        if k.startswith("grad/"):
            gr, ref = sd[k[5:]].grad, g[k]
            if ref.ndim == 0:
                assert abs(gr.norm().item() - float(ref)) <= 1e-9 * max(1.0, float(ref)), k
            else:
                np.testing.assert_allclose(gr.numpy(), ref, atol=1e-10, rtol=0, err_msg=k)


@pytest.mark.parametrize("cname,B,nv,na", [("C1", 2, 10, 0), ("C2a", 2, 15, 10), ("C3", 2, 15, 10)])
def test_named_configs_fp32(cname, B, nv, na):
    """fp32 oracle vs fp32 reference at the real model sizes: logit slices,
    per-tensor statistics and gradient norms (weights regenerated from the seed)."""
    g = np.load(os.path.join(H.GOLDEN, "%s_rec_summary.npz" % cname))
    cfg = named_config(cname)
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    for t in sd.values():
        t.requires_grad_(True)
    times = inp["times"].clone().requires_grad_(True)
    cls, feats = O.forward(sd, cfg, inp["visual"], inp["audio"], times, nv, na)
    outs = H.named_outputs(cls, feats)
    for k, v in outs.items():
        if k != "feats":
            np.testing.assert_allclose(v.detach().numpy()[:, :8], g["out/%s/slice" % k], atol=2e-5, rtol=0)
        st = g["out/%s/stats" % k]
        assert abs(v.detach().abs().max().item() - st[2]) < 1e-4
        assert abs(v.detach().double().norm().item() - st[3]) < 1e-4 * max(1.0, st[3])
    R = H.cotangents(cfg, B, nv, na, outs, seed=2, dtype=torch.float32)
    loss = sum((outs[k] * R[k]).sum() for k in outs)
    loss.backward()
    for k in g.files:
        if k.startswith("grad/") and k.endswith("/stats"):
            name = k[5:-6]
            n = sd[name].grad.double().norm().item()
            assert abs(n - g[k][3]) <= 2e-4 * max(1.0, g[k][3]), (name, n, g[k][3])


def test_param_count_c2a():
    import json
    keys = json.load(open(os.path.join(H.GOLDEN, "keys_recognition.json")))
    from tim_amd import synth
    sd = synth.make_state_dict(named_config("C2a"), seed=0)
    assert sum(v.size for v in sd.values()) == keys["_count_C2a"] == 58303640


# ---------------------------------------------------------------------------------------------------------------------
# training mode: the oracle's masks= branch against the reference in .train() under recorded dropout masks
# (tests/golden/make_golden_train.py; rec encodings.py:140-153,249, transformers.py:73,102-109)
# ---------------------------------------------------------------------------------------------------------------------
def _train_masks(g):
    out = {}
    for k in g.files:
        if k.startswith("mask/"):
            shape = tuple(int(s) for s in g["shape/" + k[5:]])
            n = int(np.prod(shape))
            out[k[5:]] = torch.from_numpy(np.unpackbits(g[k])[:n].reshape(shape).copy())
    return out


def _check_train_grads(g, sd, leaves):
    for k in g.files:
        if k.startswith("grad/"):
            np.testing.assert_allclose(sd[k[5:]].grad.numpy(), g[k], atol=1e-10, rtol=0, err_msg=k)
        if k.startswith("gin/"):
            np.testing.assert_allclose(leaves[k[4:]].grad.numpy(), g[k], atol=1e-10, rtol=0, err_msg=k)


TRAIN_REC = [("audio_visual", "audio_visual", True, 4, 2), ("audio_visual", "audio_visual", False, 4, 2),
             ("visual", "visual", True, 5, 0), ("audio", "audio", True, 0, 3)]


@pytest.mark.parametrize("im,dm,vn,nv,na", TRAIN_REC)
def test_tiny_train_mode_fp64(im, dm, vn, nv, na):
    g = np.load(os.path.join(H.GOLDEN, "tiny_train_rec_%s_%s_vn%d_nv%d_na%d.npz" % (im, dm, int(vn), nv, na)))
    cfg = H.tiny_cfg("recognition", im, dm, vn)
    masks = _train_masks(g)
    # every site the reference called is in the fixture, and dropped something
    want = {"seq"} | {"l%d_%s" % (l, s) for l in range(cfg.num_layers) for s in ("attn", "drop1", "ffn", "drop2")}
    want |= {"feat_" + m for m in ("visual", "audio") if m in im}
    assert set(masks) == want
    assert all(0 < float(m.double().mean()) < 1 for m in masks.values())
    sd, inp = H.synth_torch(cfg, 3, nv, na, seed=5, dtype=torch.float64)
    for t in sd.values():
        t.requires_grad_(True)
    leaves = {k: inp[k].clone().requires_grad_(inp[k].ndim == 3) for k in ("visual", "audio", "times")}
    te = O.time_mlp(sd, leaves["times"])
    cls, feats = O.encoder(sd, cfg, leaves["visual"], leaves["audio"], te, nv, na, masks=masks)
    outs = H.named_outputs(cls, feats)
    np.testing.assert_allclose(te.detach().numpy(), g["out/te"], atol=TOL64, rtol=0)
    assert set(outs) == {k[4:] for k in g.files if k.startswith("out/")} - {"te"}
    for k, v in outs.items():
        np.testing.assert_allclose(v.detach().numpy(), g["out/" + k], atol=TOL64, rtol=0, err_msg=k)
    # the masks matter: the evaluation-mode outputs are far from the fixture
    cls_e, _ = O.encoder(sd, cfg, leaves["visual"], leaves["audio"], te, nv, na)
    k0 = "action" if "action" in outs else "audio"
    assert np.abs(H.named_outputs(cls_e, feats)[k0].detach().numpy() - g["out/" + k0]).max() > 1e-3
    R = H.cotangents(cfg, 3, nv, na, outs, seed=5, dtype=torch.float64)
    loss = sum((outs[k] * R[k]).sum() for k in outs)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-9
    _check_train_grads(g, sd, leaves)


def test_tiny_detection_forward_train_fp64():
    """det tim.py:272-337: the queries the reference drew (returned by the model), the labels it computed for them, and the
    training-mode outputs / gradients under its recorded masks."""
    g = np.load(os.path.join(H.GOLDEN, "tiny_train_det_audio_visual_visual_single.npz"))
    cfg = H.tiny_cfg("detection", "audio_visual", "visual", False, num_class=(13, 5))
    masks = _train_masks(g)
    sd, inp = H.synth_torch(cfg, 2, 0, 0, seed=6, dtype=torch.float64)
    for t in sd.values():
        t.requires_grad_(True)
    vq = torch.from_numpy(g["v_queries"])
    nq = vq.shape[1]
    assert nq == 399
    pool = O.generate_queries(0.005).double()[0]
    # every drawn query is a row of the training pool, drawn without replacement, the same draw for every window
    assert torch.equal(vq[0], vq[1])
    hits = (vq[0][:, None, :] == pool[None]).all(-1)
    assert bool((hits.sum(1) == 1).all()) and int(hits.any(0).sum()) == nq
    tg, _, ious = O.label_queries(vq, torch.from_numpy(g["gt_segments"]),
                                  torch.zeros(2, g["gt_segments"].shape[1], 1, dtype=torch.int64), 0.25, 0.9, [13])   # ctor default, det tim.py:33
    np.testing.assert_array_equal(tg.numpy(), g["v_offsets"])
    np.testing.assert_allclose(ious.numpy(), g["v_ious"], atol=1e-15, rtol=0)
    times = torch.cat([inp["times"], vq], 1)
    vis = inp["visual"].clone().requires_grad_(True)
    aud = inp["audio"].clone().requires_grad_(True)
    cls, feats, reg = O.forward(sd, cfg, vis, aud, times, nq, 0, masks=masks)
    outs = H.named_outputs(cls, feats, reg)
    assert set(outs) == {k[4:] for k in g.files if k.startswith("out/")}
    for k, v in outs.items():
        np.testing.assert_allclose(v.detach().numpy(), g["out/" + k], atol=TOL64, rtol=0, err_msg=k)
    R = H.cotangents(cfg, 2, 0, 0, outs, seed=6, dtype=torch.float64)
    loss = sum((outs[k] * R[k]).sum() for k in outs)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-9
    _check_train_grads(g, sd, {"visual": vis, "audio": aud})
