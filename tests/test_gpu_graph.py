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


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
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
