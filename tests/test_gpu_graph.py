"""Whole-step HIP graph (tim_amd/graph.py): a replayed step must be the eager step - same logits, same gradients, weights
picked up after an optimizer update, new inputs through the static buffers - and dropout must stay dropout: the masks of a
replay come from the device-side salt (timhip_dropout_salt), so they can be reproduced by setting the salt and differ
from one replay to the next."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import helpers as H  # noqa: E402
from tim_amd import functional as F  # noqa: E402
from tim_amd.graph import GraphedStep  # noqa: E402
from tim_amd.tim import TIM  # noqa: E402

DEV = "cuda:0"
K2 = 0xD1B54A32D192ED03  # per-step multiplier of the plain (launch-argument) seeds, functional.Runtime.next_seed


def _i64(u):
    u &= (1 << 64) - 1
    return u - (1 << 64) if u >= (1 << 63) else u


def _model(cfg, sd, prec, drop):
    m = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim, feat_drop=drop,
            seq_drop=drop, d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers, enc_dropout=drop,
            num_feats=cfg.num_feats, precision=prec)
    m.load_state_dict(sd)
    return m.to(DEV).train()


def _step(model, inp, nv, na, R):
    def fn():
        for p in model.parameters():
            p.grad = None
        te = model(inp["times"], "time_mlp")
        heads, feats = model([inp["visual"], inp["audio"]], "encoder", te, nv, na)
        outs = [t for t in heads if t is not None] + [feats]
        if not R:
            g = torch.Generator().manual_seed(1)
            R.extend(torch.randn(o.shape, generator=g).to(DEV) * 0.1 for o in outs)
        torch.autograd.backward(outs, R)
        return outs
    return fn


def _snap(model, outs):
    return [o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _same(a, b, tol):
    (oa, ga), (ob, gb) = a, b
    for x, y in zip(oa, ob):
        assert torch.equal(x, y) if tol == 0 else torch.allclose(x, y, rtol=tol, atol=tol)
    assert ga.keys() == gb.keys()
    for k in ga:
        s = gb[k].abs().max().item() + 1e-12
        assert (ga[k] - gb[k]).abs().max().item() <= 2e-5 * s, k   # a few fp32 atomics (column sums) reorder between runs


@pytest.fixture
def salt_off():
    yield
    F.graph_safe_dropout(DEV, enable=False)


@pytest.mark.parametrize("prec", ["bf16", "fp32", "fp16"])
def test_replay_is_the_eager_step(prec, salt_off):
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    B, nv, na = 4, 4, 2
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=3, dtype=torch.float32)
    _, inp2 = H.synth_torch(cfg, B, nv, na, seed=9, dtype=torch.float32)
    model = _model(cfg, sd, prec, 0.0)
    static = {k: v.to(DEV).clone() for k, v in inp.items()}
    R = []
    fn = _step(model, static, nv, na, R)
    eager1 = _snap(model, fn())
    gs = GraphedStep(model, fn)
    _same(_snap(model, gs()), eager1, 0)
    # new inputs through the static buffers
    for k in static:
        static[k].copy_(inp2[k].to(DEV))
    rep2 = _snap(model, gs())
    eager2 = _snap(model, fn())
    _same(rep2, eager2, 0)
    assert not torch.equal(rep2[0][0], eager1[0][0])
    # an in-place weight update between replays reaches the captured step (its head re-casts the operand copies)
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
    rep3 = _snap(model, gs())
    eager3 = _snap(model, fn())
    _same(rep3, eager3, 0)
    assert not torch.equal(rep3[0][0], rep2[0][0])


def test_replayed_dropout_follows_the_salt(salt_off):
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    B, nv, na = 4, 4, 2
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=3, dtype=torch.float32)
    model = _model(cfg, sd, "bf16", 0.25)
    static = {k: v.to(DEV) for k, v in inp.items()}
    R = []
    fn = _step(model, static, nv, na, R)
    # plain launch-argument seeds: step counter 1
    model.rt.step = 0
    plain = _snap(model, fn())
    # the same masks from the device-side salt: seed*K + salt with salt = 1*K2 after the step's own increment
    word = F.graph_safe_dropout(DEV)
    word.fill_(_i64(K2 - F._SALT_STEP))
    salted = _snap(model, fn())
    _same(salted, plain, 0)
    gs = GraphedStep(model, fn)
    word.fill_(_i64(K2 - F._SALT_STEP))
    rep = _snap(model, gs())
    _same(rep, plain, 0)
    # the next replay advances the salt on the device: fresh masks, still a consistent forward/backward pair
    rep_b = _snap(model, gs())
    assert not torch.equal(rep_b[0][0], rep[0][0])
    word.fill_(_i64(2 * K2 - F._SALT_STEP))
    rep_c = _snap(model, gs())
    model.rt.step = 1
    F.graph_safe_dropout(DEV, enable=False)
    plain2 = _snap(model, fn())          # plain seeds, step counter 2
    _same(rep_c, plain2, 0)


def test_backward_of_a_stale_forward_is_refused(salt_off):
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    B, nv, na = 2, 4, 2
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=3, dtype=torch.float32)
    model = _model(cfg, sd, "bf16", 0.25)
    static = {k: v.to(DEV) for k, v in inp.items()}
    F.graph_safe_dropout(DEV)
    te = model(static["times"], "time_mlp")
    heads, feats = model([static["visual"], static["audio"]], "encoder", te, nv, na)
    model([static["visual"], static["audio"]], "encoder", te.detach(), nv, na)   # advances the salt
    with pytest.raises(RuntimeError, match="graph-safe dropout"):
        feats.sum().backward()


def test_c2a_production_batch_replay_is_the_eager_step_fp16(salt_off):
    """WHERE THE `graph_replay` NUMBERS OF THE BENCH LINE ARE QUOTED: C2a, B = 64 windows, fp16, .train() with the reference's
    dropout rates - the production kernels (loader-wave / tile-walk NT GEMMs, grouped weight gradients, chained layers, fused
    attention backward).  A replay and an eager step under the same dropout salt: the same masks at every site, logits and
    every gradient equal up to the order of fp32 atomic adds (at these shapes the heads' split products and the bias /
    LayerNorm column sums accumulate with atomics, so two EAGER runs are not bit-identical either: 1e-5 relative; a single
    differing keep-bit would move logits by 1e-2)."""
    from tests.test_gpu_parity import build
    from tim_amd.config import named_config
    cfg = named_config("C2a")
    B, nv, na = 64, 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=2, dtype=torch.float32)
    model = build(cfg, "fp16", sd).train()
    static = {k: v.to(DEV).clone() for k, v in inp.items()}
    R = []
    fn = _step(model, static, nv, na, R)
    word = F.graph_safe_dropout(DEV)
    gs = GraphedStep(model, fn)
    word.fill_(_i64(5 * K2 - F._SALT_STEP))
    rep = _snap(model, gs())
    word.fill_(_i64(5 * K2 - F._SALT_STEP))
    eager = _snap(model, fn())
    _same(rep, eager, 1e-5)
    assert len(rep[1]) == len(list(model.parameters())) - sum(1 for n, _ in model.named_parameters() if n.startswith("drloc_mlp"))
    assert model.rt.grads_finite()
    # the next replay draws other masks (the salt advanced on the device) and is again reproducible eagerly
    rep2 = _snap(model, gs())
    assert not torch.equal(rep2[0][0], rep[0][0])
    word.fill_(_i64(5 * K2))                     # (where the word stood before that replay's own increment)
    _same(rep2, _snap(model, fn()), 1e-5)


@pytest.mark.parametrize("layers", [6, 3])
def test_c2a_paired_weight_gradients_are_the_single_layer_ones(layers, salt_off, monkeypatch):
    """round 6: at C2a B = 64 the backward hands the weight gradients of two layers to ONE launch of eight-phase tiles
    (timhip_layer_bwd_weights_pair, wgrad_p8_kernel); TIM_AMD_WGRAD_PAIR=0 is the layer-by-layer path.  Same dropout salt: same
    logits bit for bit (the forward is untouched), every gradient equal up to the summation order inside a 9920-row
    contraction - and the pair path really is taken (timhip_layer_wgrad_pair_wins)."""
    from tests.test_gpu_parity import build
    from tim_amd.config import named_config
    import dataclasses
    cfg = dataclasses.replace(named_config("C2a"), num_layers=layers)   # (3: an odd stack - layer 0 goes out on its own)
    B, nv, na = 64, 15, 10
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=4, dtype=torch.float32)
    model = build(cfg, "fp16", sd).train()
    static = {k: v.to(DEV).clone() for k, v in inp.items()}
    R = []
    fn = _step(model, static, nv, na, R)
    names, real_call = [], F.call
    monkeypatch.setattr(F, "call", lambda name, *a: (names.append(name), real_call(name, *a))[1])
    word = F.graph_safe_dropout(DEV)
    word.fill_(_i64(7 * K2 - F._SALT_STEP))
    paired = _snap(model, fn())
    n_pair = names.count("timhip_layer_bwd_weights_pair")
    assert n_pair == cfg.num_layers // 2 and "timhip_layer_bwd_split" not in names
    assert names.count("timhip_layer_bwd_weights") == cfg.num_layers % 2
    del names[:]
    monkeypatch.setenv("TIM_AMD_WGRAD_PAIR", "0")
    word.fill_(_i64(7 * K2 - F._SALT_STEP))
    single = _snap(model, fn())
    assert "timhip_layer_bwd_weights_pair" not in names and names.count("timhip_layer_bwd_split") == cfg.num_layers
    for x, y in zip(paired[0], single[0]):
        assert torch.equal(x, y)
    assert paired[1].keys() == single[1].keys()
    for k in paired[1]:
        s_ = single[1][k].abs().max().item() + 1e-12
        assert (paired[1][k] - single[1][k]).abs().max().item() <= 1e-4 * s_, k


def test_detection_training_step_replay_is_the_eager_step(salt_off, monkeypatch):
    """bench.py's `c4_train` graph replay: the detection TRAINING step (det tim.py:272-337 + scripts/train.py:212-349) captured
    whole - query draw on the device (torch.randperm under capture, detection.py), on-device IoU labelling, encoder, focal +
    DIoU losses, the EMA normaliser, backward.  A replay must equal an eager step fed THE SAME permutation and dropout
    salt: drawn queries, regression targets, smoothed labels, IoUs, logits, loss, every gradient; consecutive replays draw
    different queries and advance the EMA normaliser on the device."""
    import bench
    from tim_amd.config import named_config
    cfg = named_config("C4")
    cfg.num_layers = 2                                   # (two layers keep the eager reference run short; shapes are C4's)
    B = 4
    model, _ = bench.build_model(cfg, "fp16", torch.device(DEV), seed=0)
    model.train()
    batch = bench.make_batch(cfg, B, 0, 0, seed=100, dev=torch.device(DEV))
    R = {"target": bench.make_det_targets(cfg, B, 6, 5, torch.device(DEV))}
    word = F.graph_safe_dropout(DEV)
    gs = GraphedStep(model, lambda: bench.det_train_step(model, batch, R["target"], R))
    norm = R[("norm", 0)]
    seen = []
    for it in range(2):
        word.fill_(_i64((7 + it) * K2 - F._SALT_STEP))
        norm_before = norm.clone()
        out = gs()
        torch.cuda.synchronize()
        rep = {"loss": out["loss"].clone(), "q": out["queries"][0].clone(), "off": out["offsets"][0].clone(),
               "iou": out["ious"][0].clone(), "lab": [t.clone() for t in out["labels"][0]],
               "cls": [t.detach().clone() for t in out["output"][0] if t is not None],
               "reg": out["output"][1][0].detach().clone(),
               "grads": {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}}
        norm_after = norm.clone()
        # the permutation the replay drew: every query is a row of the training pyramid, the same rows for every window
        nq = model.num_queries
        q = rep["q"].reshape(B, nq, 2)
        assert torch.equal(q[0], q[-1])
        pool = model.train_pool[0].to(DEV)
        hit = (q[0][:, None, :] == pool[None]).all(-1)
        assert bool((hit.sum(1) == 1).all())
        sel = hit.float().argmax(1).cpu()
        assert sel.unique().numel() == nq                                   # drawn without replacement
        seen.append(sel)
        # positives exist and the EMA advanced on the device by exactly the reference's recursion (det train.py:230)
        npos = int(torch.isfinite(rep["off"][:, 0]).sum())
        assert npos > 0
        assert abs(norm_after.item() - (0.9 * norm_before.item() + 0.1 * max(npos, 1))) <= 1e-3
        # eager step with the same permutation, salt and normaliser state
        rest = torch.tensor([i for i in range(pool.shape[0]) if i not in set(sel.tolist())], dtype=torch.int64)
        perm = torch.cat([sel, rest])
        real_randperm = torch.randperm

        def fed(n, *a, **kw):      # the eager draw is `torch.randperm(pool size)` on the host (detection.py, det tim.py:281)
            return perm.clone() if (n == pool.shape[0] and not a and not kw) else real_randperm(n, *a, **kw)
        monkeypatch.setattr(torch, "randperm", fed)
        word.fill_(_i64((7 + it) * K2 - F._SALT_STEP))
        norm.copy_(norm_before)
        eg = bench.det_train_step(model, batch, R["target"], R)
        torch.cuda.synchronize()
        monkeypatch.undo()
        assert torch.equal(eg["queries"][0], rep["q"])
        assert torch.equal(eg["offsets"][0], rep["off"]) and torch.equal(eg["ious"][0], rep["iou"])
        for a, b in zip(eg["labels"][0], rep["lab"]):
            assert torch.equal(a, b)
        for a, b in zip([t for t in eg["output"][0] if t is not None], rep["cls"]):
            assert torch.allclose(a.detach(), b, rtol=1e-5, atol=1e-5)      # (fp32 atomics reorder between any two runs)
        assert torch.allclose(eg["output"][1][0].detach(), rep["reg"], rtol=1e-5, atol=1e-6)
        assert abs(eg["loss"].item() - rep["loss"].item()) <= 1e-5 * max(1.0, abs(rep["loss"].item()))
        assert abs(norm.item() - norm_after.item()) <= 1e-4
        for n, p in model.named_parameters():
            if p.grad is None:
                continue
            s = rep["grads"][n].abs().max().item() + 1e-12
            assert (p.grad - rep["grads"][n]).abs().max().item() <= 1e-4 * s, n
    assert not torch.equal(seen[0], seen[1])                                 # a fresh draw per replay


def test_nonfinite_watch_follows_the_replays(salt_off):
    """`rt.grads_finite()` under HIP-graph replay: the flag words of the captured backward passes are rewritten by every
    replay - a replay whose cotangents overflow reads False (and its gradients really are non-finite), the next healthy replay
    True again, also after earlier reads; eager passes in between are folded in."""
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    B, nv, na = 4, 4, 2
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=3, dtype=torch.float32)
    model = _model(cfg, sd, "fp16", 0.0)
    static = {k: v.to(DEV).clone() for k, v in inp.items()}
    R = []
    fn = _step(model, static, nv, na, R)
    gs = GraphedStep(model, fn)
    gs()
    assert model.rt.grads_finite() and model.rt.grads_finite()
    good = [r.clone() for r in R]
    R[0].fill_(float("inf"))                       # the captured backward reads the cotangents from these static tensors
    gs()
    finite = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    assert not finite and not model.rt.grads_finite()
    for r, g in zip(R, good):
        r.copy_(g)
    gs()
    assert model.rt.grads_finite()
    fn()                                           # an eager pass in between
    assert model.rt.grads_finite()
    # a graph whose LAST replay overflowed is dropped: its blocks must not keep reporting that overflow (round-4 advisor
    # finding: `if rt.grads_finite(): opt.step()` would skip every step after a re-capture)
    R[0].fill_(float("inf"))
    gs()
    assert not model.rt.grads_finite()
    for r, g in zip(R, good):
        r.copy_(g)
    gs.reset()
    del gs
    assert model.rt.grads_finite()
    fn()
    assert model.rt.grads_finite()
    gs2 = GraphedStep(model, fn)                   # a re-capture is watched again
    gs2()
    assert model.rt.grads_finite()
    R[0].fill_(float("inf"))
    gs2()
    assert not model.rt.grads_finite()
