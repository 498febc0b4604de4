"""CPU oracle for the TIM encoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a fresh CPU restatement (plain torch CPU tensor ops, batch-first,
structured mask, no nn.Module) of the reference's
`TIM.forward(..., "time_mlp")` / `TIM.forward(..., "encoder")` path.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it; the product package `tim_amd` never does.

Parity pin: the reference repository holds no tests or golden vectors for this
path, so the oracle is pinned by golden vectors that
`tests/golden/make_golden.py` produced by importing the reference itself in the
build container (`/root/reference`, torch 2.10 CPU, fp64 and fp32) — see
`tests/test_oracle_golden.py`.  The training branch (`masks=`: the five dropout
sites) is pinned the same way: `tests/golden/make_golden_train.py` runs the
reference in `.train()` with `torch.nn.functional.dropout` replaced by a recorder
of seeded keep-masks (4 recognition combinations + detection `forward_train`),
`test_tiny_train_mode_fp64` / `test_tiny_detection_forward_train_fp64` feed them here.  The reference's arithmetic lives in PyTorch
(`nn.MultiheadAttention`, `nn.Linear`, `nn.LayerNorm`, `F.gelu`; pinned
pytorch=1.11.0 in environment.yml:13); the published semantics of those ops are
restated here op by op.

Parameters are addressed by the reference's state_dict key names.
"""
import math

import torch
import torch.nn.functional as F_


# ----------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------
def _lin(x, w, b, rd=None):
    """nn.Linear: y = x W^T + b.  `rd` optionally rounds the GEMM operands to a
    narrower dtype first (used only to *predict* the error of the bf16 MFMA
    path; accumulation stays in the working precision)."""
    if rd is not None:
        x = x.to(rd).to(w.dtype)
        w = w.to(rd).to(x.dtype)
    y = x @ w.t()
    return y if b is None else y + b


def _ln(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim: biased variance, eps inside the sqrt, affine."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + eps) * w + b


def _gelu(x):
    """F.gelu default = exact erf form (transformers.py:107,116-120; encodings.py:23)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def _drop(x, mask, p):
    """nn.Dropout in training mode with an explicit keep-mask (1 = keep)."""
    if mask is None or p == 0.0:
        return x
    return x * mask.to(x.dtype) * (1.0 / (1.0 - p))


# ----------------------------------------------------------------------------
# time MLP  (rec tim.py:66-74, det tim.py:68-76)
# ----------------------------------------------------------------------------
def time_mlp(sd, times, rd=None):
    h = torch.relu(_lin(times, sd["time_mlp.0.weight"], sd["time_mlp.0.bias"]))
    h = torch.relu(_lin(h, sd["time_mlp.2.weight"], sd["time_mlp.2.bias"], rd))
    h = torch.relu(_lin(h, sd["time_mlp.4.weight"], sd["time_mlp.4.bias"], rd))
    return _ln(h, sd["time_mlp.6.weight"], sd["time_mlp.6.bias"])


# ----------------------------------------------------------------------------
# feature encoding / sequence assembly
# (rec encodings.py:41-75, 102-121, 181-251; det encodings.py:35-53,83-100,152-201)
# ----------------------------------------------------------------------------
def _embed(sd, name, x, masks, cfg, rd):
    p = "feature_encoding.%s_embedder." % name
    x = _drop(x, None if masks is None else masks.get("feat_" + name), cfg.feat_drop)
    h = _gelu(_lin(x, sd[p + "1.weight"], sd[p + "1.bias"], rd))
    return _ln(h, sd[p + "3.weight"], sd[p + "3.bias"])


def feature_encoding(sd, cfg, visual, audio, te, nv, na, masks=None, rd=None):
    """Returns the assembled sequence, batch-first [B, S, E] (the reference
    transposes to [S,B,E]; layout only)."""
    fe = "feature_encoding."
    nf = cfg.num_feats
    det = cfg.variant == "detection"
    B = te.shape[0]
    d = cfg.d_model

    def qrows(cls_name, qte, mod_vec=None):
        c = sd[fe + cls_name].to(te.dtype).expand(B, qte.shape[1], d)
        r = torch.cat([c, qte], -1)
        return r if mod_vec is None else r + mod_vec

    if cfg.input_modality == "audio_visual":
        vm = sd[fe + "visual_modality_encoding"]
        am = sd[fe + "audio_modality_encoding"]
        ve = torch.cat([_embed(sd, "visual", visual, masks, cfg, rd), te[:, :nf]], -1) + vm
        ae = torch.cat([_embed(sd, "audio", audio, masks, cfg, rd), te[:, nf:2 * nf]], -1) + am
        seq = [ve, ae]
        qte = te[:, 2 * nf:]
        if "visual" in cfg.data_modality and nv > 0:
            if cfg.include_verb_noun and not det:
                seq.append(qrows("visual_verb_cls", qte[:, :nv], vm))
                seq.append(qrows("visual_noun_cls", qte[:, :nv], vm))
            seq.append(qrows("visual_action_cls", qte[:, :nv], vm))
        if "audio" in cfg.data_modality and na > 0:
            # negative slice: the audio query rows are the LAST na time rows (encodings.py:242)
            seq.append(qrows("audio_action_cls", qte[:, qte.shape[1] - na:], am))
    elif cfg.input_modality == "visual":
        seq = [torch.cat([_embed(sd, "visual", visual, masks, cfg, rd), te[:, :nf]], -1)]
        qte = te[:, nf:]
        if det:
            seq.append(qrows("visual_action_cls", qte))
        else:
            if cfg.include_verb_noun:
                seq.append(qrows("verb_cls", qte))
                seq.append(qrows("noun_cls", qte))
            seq.append(qrows("action_cls", qte))
    else:
        seq = [torch.cat([_embed(sd, "audio", audio, masks, cfg, rd), te[:, :nf]], -1)]
        qte = te[:, nf:]
        seq.append(qrows("audio_action_cls" if det else "action_cls", qte))
    x = torch.cat(seq, 1)
    return _drop(x, None if masks is None else masks.get("seq"), cfg.seq_drop)


# ----------------------------------------------------------------------------
# one post-norm encoder layer under TIM's mask
# (transformers.py:92-111 over nn.MultiheadAttention; mask rec tim.py:161-166;
#  exact structured form: SURVEY.md Appendix B)
# ----------------------------------------------------------------------------
def attention_structured(q, k, v, nfeat, drop_mask=None, p=0.0, rd=None):
    """q,k,v: [B,H,S,Dh] (q NOT yet scaled).  Token i may attend to the `nfeat`
    feature tokens and to itself: M[i,j] blocked iff j >= nfeat and j != i.
    `rd`: round the probabilities to that dtype before P.V (what an MFMA kernel with
    `rd` operands does); the self term stays in working precision."""
    Dh = q.shape[-1]
    q = q * (Dh ** -0.5)  # F.multi_head_attention_forward scales q
    kf, vf = k[:, :, :nfeat], v[:, :, :nfeat]
    s = q @ kf.transpose(-1, -2)  # [B,H,S,F]
    s_self = (q[:, :, nfeat:] * k[:, :, nfeat:]).sum(-1, keepdim=True)  # [B,H,Q,1]
    pf = torch.softmax(s[:, :, :nfeat], -1)
    pq = torch.softmax(torch.cat([s[:, :, nfeat:], s_self], -1), -1)
    if drop_mask is not None and p > 0.0:
        # drop_mask: [B,H,S,F+1]; column F is the self column (unused for feature rows)
        pf = _drop(pf, drop_mask[:, :, :nfeat, :nfeat], p)
        pq = _drop(pq, drop_mask[:, :, nfeat:], p)
    if rd is not None:
        pf = pf.to(rd).to(v.dtype)
        pq = torch.cat([pq[..., :nfeat].to(rd).to(v.dtype), pq[..., nfeat:]], -1)
    of = pf @ vf
    oq = pq[..., :nfeat] @ vf + pq[..., nfeat:] * v[:, :, nfeat:]
    return torch.cat([of, oq], 2)


def encoder_layer(sd, prefix, x, nhead, nfeat, masks=None, p=0.0, rd=None, li=0):
    B, S, E = x.shape
    Dh = E // nhead
    m = (lambda k: None) if masks is None else (lambda k: masks.get("l%d_%s" % (li, k)))
    qkv = _lin(x, sd[prefix + "self_attn.in_proj_weight"], sd[prefix + "self_attn.in_proj_bias"], rd)
    q, k, v = [t.reshape(B, S, nhead, Dh).transpose(1, 2) for t in qkv.split(E, -1)]
    if rd is not None:
        q, k, v = [t.to(rd).to(x.dtype) for t in (q, k, v)]
    o = attention_structured(q, k, v, nfeat, m("attn"), p, rd).transpose(1, 2).reshape(B, S, E)
    a = _lin(o, sd[prefix + "self_attn.out_proj.weight"], sd[prefix + "self_attn.out_proj.bias"], rd)
    x = _ln(x + _drop(a, m("drop1"), p), sd[prefix + "norm1.weight"], sd[prefix + "norm1.bias"])
    h = _drop(_gelu(_lin(x, sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"], rd)), m("ffn"), p)
    f = _lin(h, sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"], rd)
    return _ln(x + _drop(f, m("drop2"), p), sd[prefix + "norm2.weight"], sd[prefix + "norm2.bias"])


# ----------------------------------------------------------------------------
# heads (rec head.py:17-38,53-69,76-81; det head.py:27-46,65-79,89-93,117-163)
# ----------------------------------------------------------------------------
def _fc(sd, name, rows, rd):
    y = _lin(rows, sd[name + ".weight"], sd[name + ".bias"], rd)
    return y.reshape(-1, y.shape[-1])


def cls_heads(sd, cfg, x, nv, na, rd=None):
    S = x.shape[1]
    det = cfg.variant == "detection"
    verb = noun = action = audio = None
    has = lambda n: ("cls_head." + n + ".weight") in sd
    if cfg.data_modality == "audio_visual":
        aud_start = S - na if na > 0 else S
        act_start = aud_start - nv
        if has("fc_visual_verb"):
            if det:
                vr = nr = x[:, act_start:aud_start]
            else:
                noun_start = act_start - nv
                verb_start = noun_start - nv
                vr, nr = x[:, verb_start:noun_start], x[:, noun_start:act_start]
            verb = _fc(sd, "cls_head.fc_visual_verb", vr, rd)
            noun = _fc(sd, "cls_head.fc_visual_noun", nr, rd)
        action = _fc(sd, "cls_head.fc_visual_action", x[:, act_start:aud_start], rd)
        audio = _fc(sd, "cls_head.fc_audio_action", x[:, aud_start:], rd)
    elif cfg.data_modality == "visual":
        act_start = S - nv
        if has("fc_visual_verb"):
            if det:
                vr = nr = x[:, act_start:]
            else:
                vr = x[:, act_start - 2 * nv:act_start - nv]
                nr = x[:, act_start - nv:act_start]
            verb = _fc(sd, "cls_head.fc_visual_verb", vr, rd)
            noun = _fc(sd, "cls_head.fc_visual_noun", nr, rd)
        action = _fc(sd, "cls_head.fc_visual_action", x[:, act_start:], rd)
    else:
        audio = _fc(sd, "cls_head.fc_audio_action", x[:, S - na:], rd)
    return verb, noun, action, audio


def reg_heads(sd, cfg, x, nv, na, rd=None):
    """Detection regression heads: sigmoid(W3 relu(W2 relu(W1 x))) (det head.py:95-163)."""
    S = x.shape[1]

    def mlp(mod, rows):
        b = "reg_head.fc_%s_action." % mod
        h = torch.relu(_lin(rows, sd[b + "0.weight"], sd[b + "0.bias"], rd))
        h = torch.relu(_lin(h, sd[b + "2.weight"], sd[b + "2.bias"], rd))
        y = torch.sigmoid(_lin(h, sd[b + "4.weight"], sd[b + "4.bias"], rd))
        return y.reshape(-1, 2)

    if cfg.data_modality == "audio_visual":
        aud_start = S - na if na > 0 else S
        return mlp("visual", x[:, aud_start - nv:aud_start]), mlp("audio", x[:, aud_start:])
    if cfg.data_modality == "visual":
        return mlp("visual", x[:, S - nv:]), None
    return None, mlp("audio", x[:, S - na:])


def drloc_mlp(sd, x):
    """rec tim.py:129-135,190-191."""
    h = torch.relu(_lin(x, sd["drloc_mlp.0.weight"], sd["drloc_mlp.0.bias"]))
    h = torch.relu(_lin(h, sd["drloc_mlp.2.weight"], sd["drloc_mlp.2.bias"]))
    return _lin(h, sd["drloc_mlp.4.weight"], sd["drloc_mlp.4.bias"]).squeeze(2)


# ----------------------------------------------------------------------------
# whole path
# ----------------------------------------------------------------------------
def encoder(sd, cfg, visual, audio, te, nv, na, masks=None, rd=None, return_layers=False):
    """`TIM.forward_encoder` (rec tim.py:147-172): returns (cls tuple, feats[, reg tuple])."""
    x = feature_encoding(sd, cfg, visual, audio, te, nv, na, masks, rd)
    stack = "backbone" if cfg.variant == "detection" else "transformer_encoder"
    layers = [x]
    for l in range(cfg.num_layers):
        x = encoder_layer(sd, "%s.layers.%d." % (stack, l), x, cfg.nhead, cfg.F,
                          masks, cfg.enc_dropout if masks is not None else 0.0, rd, l)
        layers.append(x)
    cls = cls_heads(sd, cfg, x, nv, na, rd)
    out = [cls, x[:, :cfg.F]]
    if cfg.variant == "detection":
        out.append(reg_heads(sd, cfg, x, nv, na, rd))
    if return_layers:
        out.append(layers)
    return tuple(out)


def forward(sd, cfg, visual, audio, times, nv, na, masks=None, rd=None):
    """time_mlp followed by encoder, as the loops call them
    (rec train.py:198,209-215; test.py:108,111-117)."""
    te = time_mlp(sd, times, rd)
    return encoder(sd, cfg, visual, audio, te, nv, na, masks, rd)


# ----------------------------------------------------------------------------
# detection query pyramid (det tim.py:144-155): fixed multi-scale intervals
# ----------------------------------------------------------------------------
def generate_queries(query_size):
    qs = []
    while query_size < 1.0:
        st = torch.arange(0.0, 1.0, step=query_size / 2)
        qs.append(torch.round(torch.stack([st, st + query_size], -1), decimals=3))
        query_size *= 2
    return torch.cat(qs, 0).unsqueeze(0)


# ----------------------------------------------------------------------------
# detection query labelling (det tim.py:157-270).  Pinned by tests/golden/labels_*.npz, generated from the reference's own
# TIM.label_queries by tests/golden/make_golden_r2.py.
# ----------------------------------------------------------------------------
def query_ious(queries, segs):
    """det tim.py:186-212.  queries [B,Nq,2], segs [B,Ng,2] (fp32) -> (ious [B,Nq,Ng], shifted segs [B,Ng,2]).
    Every interval of a window is shifted by |min(0, earliest ground-truth start)| first (:196-203); the reference does that
    in place on its expanded copies, so the segments it later hands back as regression targets are the SHIFTED ones."""
    off = torch.abs(torch.clamp(segs[:, :, 0].min(dim=-1)[0], max=0.0))[:, None]        # [B,1]
    q_s, q_e = queries[:, :, 0] + off, queries[:, :, 1] + off                             # [B,Nq]
    g_s, g_e = segs[:, :, 0] + off, segs[:, :, 1] + off                                   # [B,Ng]
    i_s = torch.maximum(q_s[:, :, None], g_s[:, None, :])
    i_e = torch.minimum(q_e[:, :, None], g_e[:, None, :])
    inter = torch.clamp(i_e - i_s, min=0.0)
    unions = (g_e - g_s)[:, None, :] + (q_e - q_s)[:, :, None] - inter
    return inter / unions, torch.stack([g_s, g_e], -1)


def smooth_one_hot(idx, n, ls):
    """(F.one_hot(idx, n+1) * ls + (1-ls)/(n+1))[:, :-1] as det tim.py:170-183 evaluates it: the int64 one-hot times a Python
    float is an fp32 tensor, the second Python float is rounded to fp32 before the add."""
    on = torch.tensor(float(ls), dtype=torch.float32)
    base = torch.tensor((1.0 - float(ls)) / (n + 1), dtype=torch.float32)
    out = base.expand(idx.shape[0], n).clone()
    rows = torch.nonzero(idx < n).flatten()
    out[rows, idx[rows]] = on + base
    return out


def label_queries(queries, segs, gt_labels, iou_threshold, label_smoothing, num_classes):
    """det tim.py:214-270.  queries [B,Nq,2] fp32; segs [B,Ng,2] fp32; gt_labels [B,Ng,NL] int64 (-1 = padding);
    num_classes: NL class counts.  Returns (targets [B*Nq,2] - inf for negatives, [smoothed label matrix per label column],
    ious [B*Nq]).  The first maximum wins ties (torch.argmax)."""
    B, Nq = queries.shape[:2]
    ious, shifted = query_ious(queries, segs)
    idx = ious.argmax(-1)                                                   # [B,Nq]
    best = torch.gather(ious, 2, idx[..., None]).squeeze(-1)
    tg = torch.gather(shifted[:, None].expand(-1, Nq, -1, -1), 2, idx[..., None, None].expand(-1, -1, 1, 2)).squeeze(2).clone()
    lab = torch.gather(gt_labels[:, None].expand(-1, Nq, -1, -1), 2,
                       idx[..., None, None].expand(-1, -1, 1, gt_labels.shape[-1])).squeeze(2).clone()
    neg = best < iou_threshold
    tg[neg] = float("inf")
    lab[neg] = -1
    lab = lab.reshape(B * Nq, -1)
    mats = []
    for c, n in enumerate(num_classes):
        col = torch.where(lab[:, c] == -1, torch.full_like(lab[:, c], n), lab[:, c])     # "no object" -> the dropped column n
        mats.append(smooth_one_hot(col, n, label_smoothing))
    return tg.reshape(B * Nq, 2), mats, best.reshape(-1)


def to_torch(sd_np, dtype=torch.float32):
    return {k: torch.from_numpy(v).to(dtype) for k, v in sd_np.items()}


# ---------------------------------------------------------------------------------------------------------------------
# loss tail.  Pinned by tests/golden/loss_*.npz, generated from the reference's own mixup_criterion /
# drloc functions by tests/golden/make_golden_loss.py.
# ---------------------------------------------------------------------------------------------------------------------
def mixup_ce(logits, ya, yb, lam, smoothing=0.2):
    """scripts/train.py:46-49,218-258 + utils/mixup.py:24-39: rows with target -1 are dropped on each side, the
    label-smoothed CE is averaged over the kept rows, the two sides are blended with lam."""
    crit = torch.nn.CrossEntropyLoss(label_smoothing=smoothing, ignore_index=-1)
    va = ya != -1
    loss_a = crit(logits[va], ya[va]).mean()
    if yb is None:
        return loss_a
    vb = yb != -1
    loss_b = crit(logits[vb], yb[vb]).mean()
    return lam * loss_a + (1 - lam) * loss_b


def drloc_loss(sd, x1, x2, pos_1, pos_2):
    """models/helpers/losses/drloc.py:10-41 with the sampled positions given: x1, x2 [n, l, D]; pos [n, m]"""
    n, l, D = x1.shape
    idx = lambda x, pos: torch.gather(x, 1, pos.to(x.device).long().unsqueeze(-1).expand(-1, -1, D))   # collect_samples
    pts = torch.cat([idx(x1, pos_1), idx(x2, pos_2)], dim=2)
    deltax = torch.abs((pos_1 - pos_2).float())   # fp32, as drloc.py:21-22 (the division rounds in fp32)
    deltax /= l
    return torch.nn.functional.l1_loss(deltax, drloc_mlp(sd, pts))


def focal_loss(x, t, weights=None, alpha=0.25, gamma=2.0, reduction="sum"):
    """detection models/helpers/losses/sigmoid.py:5-52 under losses/loss.py:5-14 (row weights, reduction)"""
    p = torch.sigmoid(x)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(x, t, reduction="none")
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    if weights is not None:
        loss = loss * weights[:, None]
    return loss.sum() if reduction == "sum" else (loss.mean() if reduction == "mean" else loss)


def diou_1d(pred, tgt, eps=1e-8):
    """detection models/helpers/losses/iou.py:4-65 ("sum").  The reference function is TorchScript; once compiled its
    autodiff gives min / max a gradient only under STRICT inequality and clamp(min) only where the input >= min (its first,
    profiling calls split exact ties like eager torch) - the torch.where forms below are the compiled behaviour; the golden
    vectors contain no exact ties."""
    lp, rp, lg, rg = pred[:, 0], pred[:, 1], tgt[:, 0].detach(), tgt[:, 1].detach()
    smin = lambda a, b: torch.where(a < b, a, b)
    smax = lambda a, b: torch.where(a > b, a, b)
    sclamp = lambda u: torch.where(u >= eps, u, torch.full_like(u, eps).detach())
    inter = smin(rp, rg) + smin(lp, lg)
    union = (lp + rp) + (lg + rg) - inter
    iou = inter / sclamp(union)
    len_c = smax(lp, lg) + smax(rp, rg)
    rho = 0.5 * (rp - lp - rg + lg)
    return (1.0 - iou + torch.square(rho / sclamp(len_c))).sum()
