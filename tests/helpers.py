"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os
import re

import numpy as np
import torch

from tim_amd import synth
from tim_amd.config import named_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def tiny_cfg(variant, im, dm, vn, num_class=None):
    c = named_config("tiny")
    c.input_modality, c.data_modality, c.include_verb_noun = im, dm, vn
    c.variant = variant
    if num_class is not None:
        c.num_class = num_class
    elif not vn:
        c.num_class = [13, 5]
    return c


def rec_golden_cases():
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN, "tiny_rec_*.npz"))):
        m = re.match(r"tiny_rec_(audio_visual|visual|audio)_(audio_visual|visual|audio)_vn(\d)_nv(\d+)_na(\d+)\.npz",
                     os.path.basename(f))
        im, dm, vn, nv, na = m.group(1), m.group(2), bool(int(m.group(3))), int(m.group(4)), int(m.group(5))
        out.append((os.path.basename(f), im, dm, vn, nv, na))
    return out


DET_CASES = [("audio_visual", "visual", (13, 5), "single"),
             ("audio_visual", "audio_visual", (13, 5), "single"),
             ("audio_visual", "audio_visual", [[7, 11, 13], 5], "vn"),
             ("visual", "visual", [[7, 11, 13], 5], "vn"),
             ("audio", "audio", (13, 5), "single")]


def synth_torch(cfg, B, nv, na, seed, dtype):
    sd = {k: torch.from_numpy(v).to(dtype) for k, v in
          synth.make_state_dict(cfg, seed=seed, dtype=np.float64).items()}
    inp = {k: torch.from_numpy(v).to(dtype) for k, v in
           synth.make_inputs(cfg, B, nv, na, seed=seed, dtype=np.float64).items()}
    return sd, inp


def named_outputs(cls, feats, reg=None):
    out = {"feats": feats}
    for k, v in zip(("verb", "noun", "action", "audio"), cls):
        if v is not None:
            out[k] = v
    if reg is not None:
        for k, v in zip(("reg_visual", "reg_audio"), reg):
            if v is not None:
                out[k] = v
    return out


def cotangents(cfg, B, nv, na, outs, seed, dtype):
    R = synth.make_cotangents(cfg, B, nv, na, {k: tuple(v.shape) for k, v in outs.items()},
                              seed=seed, dtype=np.float64)
    return {k: torch.from_numpy(v).to(dtype) for k, v in R.items()}
