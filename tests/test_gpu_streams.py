"""The two stream configurations of the encoder backward - everything on the current stream (default) and the layer's
weight-gradient launch on a side stream (TIM_AMD_OVERLAP_WGRAD=1) - run the same kernels on the same data: identical
outputs and gradients, with dropout on."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import helpers as H  # noqa: E402
from tim_amd.tim import TIM  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_side_stream_weight_gradients_match_single_stream(prec):
    cfg = H.tiny_cfg("recognition", "audio_visual", "audio_visual", True)
    B, nv, na = 6, 4, 2
    sd, inp = H.synth_torch(cfg, B, nv, na, seed=3, dtype=torch.float32)
    m = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim, feat_drop=0.1,
            seq_drop=0.1, d_model=cfg.d_model, nhead=cfg.nhead, num_layers=cfg.num_layers, enc_dropout=0.1,
            num_feats=cfg.num_feats, precision=prec)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    dinp = {k: v.to(DEV) for k, v in inp.items()}
    res = []
    for overlap in (False, True, False):
        m.rt.overlap_wgrad = overlap
        m.rt.step = 0                       # same dropout masks
        for p in m.parameters():
            p.grad = None
        te = m(dinp["times"], "time_mlp")
        heads, feats = m([dinp["visual"], dinp["audio"]], "encoder", te, nv, na)
        outs = [t for t in heads if t is not None] + [feats]
        g = torch.Generator().manual_seed(1)
        torch.autograd.backward(outs, [torch.randn(o.shape, generator=g).to(DEV) * 0.1 for o in outs])
        torch.cuda.synchronize()
        res.append(([o.detach().clone() for o in outs], {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    for (o1, g1) in res[1:]:
        for a, b in zip(res[0][0], o1):
            assert torch.equal(a, b)
        assert g1.keys() == res[0][1].keys()
        for k in g1:
            s = res[0][1][k].abs().max().item() + 1e-12
            assert (g1[k] - res[0][1][k]).abs().max().item() <= 2e-5 * s, k   # fp32 atomics in the column sums reorder
