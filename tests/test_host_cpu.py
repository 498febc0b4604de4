"""CPU-only checks of the host mirror: drop-in boundary (state_dict names/shapes/order for every
modality combo), C-ABI library symbols, token-row plan vs the oracle's assembly, loud failure
without a GPU."""
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import tim_oracle as O
from tests import helpers as H
from tim_amd import _lib
from tim_amd.functional import EncoderPlan


def _build(variant, cfg):
    if variant == "detection":
        from tim_amd.detection import TIM
        return TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim,
                   d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers,
                   input_modality=cfg.input_modality, data_modality=cfg.data_modality, num_feats=cfg.num_feats,
                   include_verb_noun=cfg.include_verb_noun)
    from tim_amd.tim import TIM
    return TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim,
               d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers,
               input_modality=cfg.input_modality, data_modality=cfg.data_modality, num_feats=cfg.num_feats,
               include_verb_noun=cfg.include_verb_noun)


def test_state_dict_keys_match_reference_recognition():
    keys = json.load(open(os.path.join(H.GOLDEN, "keys_recognition.json")))
    n = 0
    for k, v in keys.items():
        if k.startswith("_"):
            continue
        im, dm, vn = k.split("/")
        m = _build("recognition", H.tiny_cfg("recognition", im, dm, bool(int(vn))))
        assert [[a, list(b.shape)] for a, b in m.state_dict().items()] == v, k
        n += 1
    assert n == 10


def test_state_dict_keys_match_reference_detection():
    keys = json.load(open(os.path.join(H.GOLDEN, "keys_detection.json")))
    for im, dm, nc, tag in H.DET_CASES:
        m = _build("detection", H.tiny_cfg("detection", im, dm, tag == "vn", num_class=nc))
        assert [[a, list(b.shape)] for a, b in m.state_dict().items()] == keys["%s/%s/%s" % (im, dm, tag)]
        assert m.num_queries == 399 and m.train_pool.shape[1] == 799  # det tim.py:140-142


def test_full_size_parameter_count_and_fresh_layers_identical():
    from tim_amd.tim import TIM
    m = TIM([[97, 300, 3806], 44])
    assert sum(p.numel() for p in m.parameters()) == 58303640
    l0, l5 = m.transformer_encoder.layers[0], m.transformer_encoder.layers[5]
    assert torch.equal(l0.linear1.weight, l5.linear1.weight)  # _get_clones deepcopy, transformers.py:113-114


def test_load_state_dict_roundtrip_with_reference_named_weights():
    from tim_amd import synth
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=3).items()}
    m = _build("recognition", cfg)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k])


def test_header_is_plain_c(tmp_path):
    """include/timhip.h is the drop-in boundary: it must compile as C99 (and C++) on its own, no torch / HIP types"""
    import shutil
    import subprocess
    inc = os.path.join(os.path.dirname(H.GOLDEN), "..", "include")
    src = tmp_path / "t.c"
    src.write_text('#include "timhip.h"\nint main(void) { TimDesc d; (void)d; return TIMHIP_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    if shutil.which("g++"):
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", inc, "-fsyntax-only", "-x", "c++", str(src)])
    hdr = open(os.path.join(inc, "timhip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)            # comments may cite torch / HIP names
    assert "#include <hip" not in code and "torch" not in code and "hipStream_t" not in code and "at::" not in code


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(os.path.dirname(H.GOLDEN), "..", "include", "timhip.h")).read()
    declared = set(re.findall(r"\b(timhip_[a-z0-9_]+)\s*\(", hdr))
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.exported_symbols())
    assert lib.timhip_version() == _lib.ABI_VERSION == 6
    assert lib.timhip_strerror(-3).decode().startswith("workspace")


def test_cpu_tensors_fail_loudly():
    cfg = H.tiny_cfg("recognition", "visual", "visual", True)
    m = _build("recognition", cfg)
    with pytest.raises(_lib.TimHipError):
        m(torch.zeros(2, 11, 2), "time_mlp")
    with pytest.raises(_lib.TimHipError):
        m([torch.zeros(2, 6, 24), torch.zeros(2, 0)], "encoder", torch.zeros(2, 11, 32), 5, 0)


@pytest.mark.parametrize("fname,im,dm,vn,nv,na", H.rec_golden_cases())
def test_token_row_plan_matches_oracle_assembly(fname, im, dm, vn, nv, na):
    """Evaluate the row table on CPU and compare with the oracle's feature_encoding()."""
    cfg = H.tiny_cfg("recognition", im, dm, vn)
    sd, inp = H.synth_torch(cfg, 3, nv, na, seed=1, dtype=torch.float64)
    te = O.time_mlp(sd, inp["times"])
    want = O.feature_encoding(sd, cfg, inp["visual"], inp["audio"], te, nv, na)
    plan = EncoderPlan(cfg, te.shape[1], nv, na)
    assert plan.S == want.shape[1] == cfg.F + cfg.num_queries(nv, na)
    d = cfg.d_model
    e = {}
    for name, slot in plan.embedders:
        e[slot] = O._embed(sd, name, inp[name], None, cfg, None)
    got = torch.zeros_like(want)
    for s, (kind, src, te_row, mod) in enumerate(plan.rows):
        left = sd["feature_encoding." + plan.cls_names[src]].reshape(d) if kind == 1 else e[0 if kind == 0 else 1][:, src]
        row = torch.cat([left.expand(3, d) if kind == 1 else left, te[:, te_row]], -1)
        if mod >= 0:
            row = row + sd["feature_encoding." + plan.mod_names[mod]].reshape(-1)
        got[:, s] = row
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-12)
    # head slices agree with the oracle's tail slicing
    cls = O.cls_heads(sd, cfg, want, nv, na)
    for slot, pname, s0, n in plan.heads:
        ref = dict(zip(("verb", "noun", "action", "audio"), cls))[slot]
        w, b = sd["cls_head." + pname + ".weight"], sd["cls_head." + pname + ".bias"]
        mine = (want[:, s0:s0 + n] @ w.t() + b).reshape(-1, w.shape[0])
        np.testing.assert_allclose(mine.numpy(), ref.numpy(), atol=1e-12)


def test_avga_pooling_prestep_matches_reference():
    """pool_features=True (AVE recipe): same state_dict keys as the reference model, and the pooling module - stock torch, an
    input pre-step outside the HIP path - reproduces the reference module's output (tests/golden/make_golden_r2.py)"""
    from tim_amd.tim import TIM
    g = np.load(os.path.join(H.GOLDEN, "avga_tiny.npz"))
    m = TIM([[7, 11, 13], 5], visual_input_dim=24, audio_input_dim=40, d_model=32, nhead=2, num_layers=2, num_feats=6,
            pool_features=True)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    m.pool.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("pool/")})
    out = m.pool(torch.from_numpy(g["audio"]), torch.from_numpy(g["video"]))
    assert (out - torch.from_numpy(g["pooled"])).abs().max().item() <= 1e-6


def test_constructor_rejects_shapes_the_kernels_do_not_cover():
    from tim_amd.tim import TIM
    with pytest.raises(ValueError, match="feature tokens"):
        TIM([[7, 11, 13], 5], d_model=32, nhead=2, num_layers=1, num_feats=100)      # 200 keys per head > 192
    TIM([[7, 11, 13], 5], d_model=32, nhead=2, num_layers=1, num_feats=96)


def test_workspaces_captured_by_a_graph_are_retired_not_freed():
    from tim_amd.tim import TIM
    m = TIM([[7, 11, 13], 5], visual_input_dim=24, audio_input_dim=40, d_model=32, nhead=2, num_layers=1, num_feats=6)
    dev = torch.device("cpu")
    a = m._workspace(1000, dev)
    assert m._workspace(500, dev) is a
    m._ws_pinned = True                      # what GraphedStep sets
    b = m._workspace(5000, dev)
    assert b is not a and any(w is a for w in m._ws_retired)      # the captured buffer stays alive
    st = m.dropout_rng_state()
    m.rt.step = 17
    m.set_dropout_rng_state(st)
    assert m.rt.step == st["dropout_step"]


def test_build_model_takes_the_reference_parsers_namespaces():
    """`build_model(args)` with exactly the attributes the reference's two argument parsers produce
    (recognition/time_interval_machine/utils/parser.py:48-70, detection/.../utils/parser.py:28-50): no `variant` attribute -
    the detection namespace is recognised by its own fields (--iou_threshold, the `feedfoward_scale` spelling)"""
    import argparse
    from tim_amd.build import build_model
    from tim_amd.detection import TIM as DetTIM
    from tim_amd.tim import TIM as RecTIM
    common = dict(num_gpus=0, workers=4, visual_input_dim=24, audio_input_dim=40, feat_dropout=0.5, seq_dropout=0.5, d_model=32,
                  nhead=2, num_layers=2, enc_dropout=0.1, model_modality="audio_visual", data_modality="audio_visual",
                  num_feats=6, include_verb_noun=True)
    rec = argparse.Namespace(num_class=[[7, 11, 13], 5], feedforward_scale=4, apply_feature_pooling=False, **common)
    m, _ = build_model(rec)
    assert type(m) is RecTIM and m.dim_feedforward == 128
    det = argparse.Namespace(num_class=[[7, 11, 13], 5], feedfoward_scale=4, iou_threshold=0.6, label_smoothing=0.9, **common)
    m, _ = build_model(det)
    assert type(m) is DetTIM and hasattr(m, "reg_head")
    rec.variant = "recognition"      # an explicit attribute still wins
    assert type(build_model(rec)[0]) is RecTIM


def test_layer_split_mode_is_validated(monkeypatch):
    """Runtime.layer_split (opt-in fp16 margin mode): unknown names and unknown TIM_AMD_SPLIT_LAYER_WEIGHTS values raise, widths
    the wrapped-operand product cannot take (E or FF not a multiple of 64) fall back to plain weights with a warning, and
    `split_outproj` follows an assignment made after construction."""
    import warnings
    from tim_amd.functional import Runtime
    rt = Runtime("fp16")
    assert rt.layer_split == () and not rt.split_outproj
    rt.layer_split = ("out",)
    assert rt.split_outproj and rt.layer_split_for(1024, 2048) == ("out",)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert rt.layer_split_for(96, 192) == () and rt.layer_split_flags(96, 192) == 0
        assert rt.layer_split_for(160, 320) == ()
    assert len(w) == 1 and "plain" in str(w[0].message)          # warned once
    with pytest.raises(ValueError):
        rt.layer_split = ("out", "ffn")
    with pytest.raises(ValueError):
        Runtime("bf16").layer_split = ("out",)
    monkeypatch.setenv("TIM_AMD_SPLIT_LAYER_WEIGHTS", "everything")
    with pytest.raises(ValueError):
        Runtime("fp16")
    monkeypatch.setenv("TIM_AMD_SPLIT_LAYER_WEIGHTS", "all")
    assert Runtime("fp16").layer_split == ("in", "out", "l1", "l2")
    assert Runtime("bf16").layer_split == ()                     # (the switch is an fp16-mode option)


def test_gradient_bucket_layout_is_cached_and_consistent():
    """tim.py:_BucketLayout (round 5: computed once per model instead of per backward pass): every parameter's view has its shape
    and lies inside its bucket, buckets are contiguous in completion order, the pieces of the one split call tile the allocation,
    and the zero fills expressed through those pieces cover exactly the ranges the layout lists (whole accumulated buckets;
    LayerNorm slices, slot paddings and alignment tails of the overwritten layer buckets)."""
    import torch
    from tim_amd.tim import TIM, _BucketLayout, _GradBuckets
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    m = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim, d_model=cfg.d_model,
            nhead=cfg.nhead, num_layers=cfg.num_layers, num_feats=cfg.num_feats)
    names, params = m._encoder_param_names, m._encoder_param_list()
    for overwrite in (False, True):
        lay = _BucketLayout(names, params, m._bucket_of, overwrite)
        assert sum(lay.piece_sizes) == lay.total and lay.order[0] == "heads" and lay.order[-1] == "front"
        starts, pos = [], 0
        for sz in lay.piece_sizes:
            starts.append(pos)
            pos += sz
        got = set()
        for kind, k, rng in lay.zero_spec:
            if kind == "flat":
                got.add(lay.bucket[k])
            elif kind == "piece":
                got.add((starts[k], lay.piece_sizes[k]))
            else:
                got.add((starts[k] + rng[0], rng[1] - rng[0]))
        assert got == set(lay.zero)
        gb = m._alloc_grad_buckets(names, params, torch.device("cpu"), layer_overwrite=overwrite)
        assert m._alloc_grad_buckets(names, params, torch.device("cpu"), layer_overwrite=overwrite).base.numel() == gb.base.numel()
        assert len(m._bucket_layouts) <= 2                       # one layout per overwrite mode, reused
        ref = _GradBuckets(m.rt, names, params, torch.device("cpu"), m._bucket_of, overwrite)   # (a fresh layout: same result)
        for n, p in zip(names, params):
            v = gb.views[n]
            assert v.shape == p.shape and v.is_contiguous()
            off = (v.data_ptr() - gb.base.data_ptr()) // 4
            st, nb = lay.bucket[m._bucket_of(n)]
            assert st <= off and off + p.numel() <= st + nb
            assert (ref.views[n].data_ptr() - ref.base.data_ptr()) // 4 == off
        # consecutive buckets are contiguous
        for a, b in zip(lay.order, lay.order[1:]):
            assert lay.bucket[a][0] + lay.bucket[a][1] == lay.bucket[b][0]
    # a parameter replaced under the same name with another size (a resized head) must not meet the cached layout
    i = names.index("cls_head.fc_visual_verb.weight")
    params2 = list(params)
    params2[i] = torch.nn.Parameter(torch.zeros(params[i].shape[0] + 3, params[i].shape[1]))
    gb2 = m._alloc_grad_buckets(names, params2, torch.device("cpu"), layer_overwrite=True)
    assert gb2.views[names[i]].shape == params2[i].shape
    assert gb2.base.numel() >= gb.base.numel()
