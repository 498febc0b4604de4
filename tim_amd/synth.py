"""Deterministic synthetic weights and feature windows.

Everything is derived from a splitmix64 counter hash so that the same tensors
can be regenerated bit-for-bit in this container (to feed the imported
reference when golden vectors are made) and on the GPU box (to feed the HIP
path and the oracle) without shipping any weight file and without depending on
a library RNG stream.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    b = np.array([(seed * 0x100000001B3 + h * 0x9E3779B1 + 0x1234567) & 0xFFFFFFFFFFFFFFFF],
                 dtype=np.uint64)
    return _splitmix64(_splitmix64(b))[0]


def uniform01(seed: int, name: str, shape) -> np.ndarray:
    """float64 uniforms in (0,1), one independent stream per (seed, name)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        base = _stream_base(seed, name)
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base) & _M64
        bits = _splitmix64(ctr) >> np.uint64(11)  # 53 random bits
    u = (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def uniform(seed, name, shape, lo, hi):
    return lo + (hi - lo) * uniform01(seed, name, shape)


def normal(seed, name, shape, std=1.0, mean=0.0):
    u1 = uniform01(seed, name + "/u1", shape)
    u2 = uniform01(seed, name + "/u2", shape)
    return mean + std * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# --------------------------------------------------------------------------
# state_dict in the reference's key names (SURVEY.md section 8a "Parameter inventory")
# --------------------------------------------------------------------------
def _linear(sd, seed, name, out_f, in_f, bias_const=None):
    k = 1.0 / np.sqrt(in_f)
    sd[name + ".weight"] = uniform(seed, name + ".weight", (out_f, in_f), -k, k)
    if bias_const is None:
        sd[name + ".bias"] = uniform(seed, name + ".bias", (out_f,), -k, k)
    else:
        sd[name + ".bias"] = np.full((out_f,), bias_const) + normal(seed, name + ".bias", (out_f,), 0.02)


def _layernorm(sd, seed, name, n):
    sd[name + ".weight"] = 1.0 + normal(seed, name + ".weight", (n,), 0.02)
    sd[name + ".bias"] = normal(seed, name + ".bias", (n,), 0.02)


def param_shapes(cfg):
    """Ordered {name: shape} of the reference module's state_dict for `cfg`."""
    return {k: v.shape for k, v in make_state_dict(cfg, seed=0, _shapes_only=True).items()}


def make_state_dict(cfg, seed=0, dtype=np.float32, _shapes_only=False):
    """Synthetic weights, independent per layer (a freshly constructed reference
    model has L identical layers because of deepcopy, transformers.py:113-114;
    distinct layers make the parity test sensitive to layer ordering)."""
    d, E, FF = cfg.d_model, cfg.E, cfg.FF
    det = cfg.variant == "detection"
    sd = {}
    _linear(sd, seed, "time_mlp.0", d, 2)
    _linear(sd, seed, "time_mlp.2", d, d)
    _linear(sd, seed, "time_mlp.4", d, d)
    _layernorm(sd, seed, "time_mlp.6", d)

    fe = "feature_encoding."
    cls_std = 0.01 if not _shapes_only else 0.01

    def small(name, shape):
        # reference init is N(0, 0.01) (encodings.py:30,158-175); use 0.5 so the
        # parity test is sensitive to these terms.
        sd[name] = normal(seed, name, shape, 0.5)

    av = cfg.input_modality == "audio_visual"
    if av or cfg.input_modality == "visual":
        _linear(sd, seed, fe + "visual_embedder.1", d, cfg.visual_input_dim)
        _layernorm(sd, seed, fe + "visual_embedder.3", d)
    if av or cfg.input_modality == "audio":
        _linear(sd, seed, fe + "audio_embedder.1", d, cfg.audio_input_dim)
        _layernorm(sd, seed, fe + "audio_embedder.3", d)
    if av:
        small(fe + "visual_modality_encoding", (1, 1, E))
        small(fe + "audio_modality_encoding", (1, 1, E))
        if "visual" in cfg.data_modality:
            small(fe + "visual_action_cls", (1, 1, d))
            if cfg.include_verb_noun and not det:
                small(fe + "visual_verb_cls", (1, 1, d))
                small(fe + "visual_noun_cls", (1, 1, d))
        if "audio" in cfg.data_modality:
            small(fe + "audio_action_cls", (1, 1, d))
    elif cfg.input_modality == "visual":
        if det:
            small(fe + "visual_action_cls", (1, 1, d))
        else:
            small(fe + "action_cls", (1, 1, d))
            if cfg.include_verb_noun:
                small(fe + "verb_cls", (1, 1, d))
                small(fe + "noun_cls", (1, 1, d))
    else:
        small(fe + ("audio_action_cls" if det else "action_cls"), (1, 1, d))

    # classification heads (rec head.py:4-81, det head.py:7-93)
    nc = cfg.num_class
    if cfg.data_modality == "audio_visual":
        vcls, acls = nc[0], nc[1]
        vn_head = isinstance(nc, list) if det else isinstance(nc[0], list)
        if det and vn_head:
            vcls = nc[0]
    elif cfg.data_modality == "visual":
        vcls, acls = nc[0], None
        vn_head = isinstance(vcls, list)
    else:
        vcls, acls = None, nc[1]
        vn_head = False
    if vcls is not None:
        if vn_head:
            _linear(sd, seed, "cls_head.fc_visual_verb", vcls[0], E)
            _linear(sd, seed, "cls_head.fc_visual_noun", vcls[1], E)
            _linear(sd, seed, "cls_head.fc_visual_action", vcls[2], E)
        else:
            _linear(sd, seed, "cls_head.fc_visual_action", vcls, E)
    if acls is not None:
        _linear(sd, seed, "cls_head.fc_audio_action", acls, E)
    if det:
        for mod, present in (("visual", vcls is not None), ("audio", acls is not None)):
            if present:
                base = "reg_head.fc_%s_action." % mod
                _linear(sd, seed, base + "0", E // 2, E)
                _linear(sd, seed, base + "2", E // 2, E // 2)
                _linear(sd, seed, base + "4", 2, E // 2)

    stack = "backbone" if det else "transformer_encoder"
    for l in range(cfg.num_layers):
        p = "%s.layers.%d." % (stack, l)
        k = 1.0 / np.sqrt(E)
        sd[p + "self_attn.in_proj_weight"] = uniform(seed, p + "in_w", (3 * E, E), -k, k)
        sd[p + "self_attn.in_proj_bias"] = uniform(seed, p + "in_b", (3 * E,), -k, k)
        _linear(sd, seed, p + "self_attn.out_proj", E, E)
        _linear(sd, seed, p + "linear1", FF, E)
        _linear(sd, seed, p + "linear2", E, FF)
        _layernorm(sd, seed, p + "norm1", E)
        _layernorm(sd, seed, p + "norm2", E)

    _linear(sd, seed, "drloc_mlp.0", d, 4 * d)
    _linear(sd, seed, "drloc_mlp.2", d, d)
    _linear(sd, seed, "drloc_mlp.4", 1, d)
    return {k: np.ascontiguousarray(v, dtype=dtype) for k, v in sd.items()}


# --------------------------------------------------------------------------
# synthetic feature windows (SURVEY.md section 8d "Synthetic inputs")
# --------------------------------------------------------------------------
def make_inputs(cfg, batch, nv, na, seed=0, dtype=np.float32):
    """Returns dict(visual [B,nf,Cv], audio [B,nf,Ca], times [B,T,2]).

    times row order: vis feats, aud feats, v queries, a queries
    (rec sliding_window.py:402).  A missing modality is a [B,0] tensor
    (sliding_window.py:352-353).
    """
    nf = cfg.num_feats
    av = cfg.input_modality == "audio_visual"
    out = {}
    has_v = av or cfg.input_modality == "visual"
    has_a = av or cfg.input_modality == "audio"
    out["visual"] = (normal(seed, "in/visual", (batch, nf, cfg.visual_input_dim))
                     if has_v else np.zeros((batch, 0)))
    out["audio"] = (normal(seed, "in/audio", (batch, nf, cfg.audio_input_dim))
                    if has_a else np.zeros((batch, 0)))
    k = np.arange(nf, dtype=np.float64) / nf
    ft = np.stack([k, k + 1.0 / nf], -1)  # [nf,2]
    rows = [np.broadcast_to(ft, (batch, nf, 2))] * (2 if av else 1)

    def q(name, n):
        st = uniform(seed, name + "/s", (batch, n), 0.0, 0.9)
        ln = uniform(seed, name + "/l", (batch, n), 0.02, 0.3)
        return np.stack([st, np.minimum(st + ln, 1.0)], -1)

    if nv > 0:
        rows.append(q("in/vq", nv))
    if na > 0:
        rows.append(q("in/aq", na))
    out["times"] = np.concatenate(rows, 1)
    return {k2: np.ascontiguousarray(v, dtype=dtype) for k2, v in out.items()}


def make_cotangents(cfg, batch, nv, na, shapes, seed=0, dtype=np.float32):
    """Fixed random R tensors for the loss L = sum <out, R>."""
    return {k: np.ascontiguousarray(normal(seed, "R/" + k, s), dtype=dtype)
            for k, s in shapes.items()}
