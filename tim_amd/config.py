"""Configuration record for one TIM encoder instance.

Field names follow the reference constructor
(recognition/time_interval_machine/models/tim.py:18-34,
 detection/time_interval_machine/models/tim.py:18-35).
"""
from dataclasses import dataclass, field
from typing import Any


@dataclass
class TimConfig:
    num_class: Any = field(default_factory=lambda: [[97, 300, 3806], 44])
    visual_input_dim: int = 1024
    audio_input_dim: int = 2304
    feat_drop: float = 0.5
    seq_drop: float = 0.5
    d_model: int = 512
    feedforward_scale: int = 4
    nhead: int = 8
    num_layers: int = 6
    enc_dropout: float = 0.1
    input_modality: str = "audio_visual"
    data_modality: str = "audio_visual"
    num_feats: int = 50
    include_verb_noun: bool = True
    variant: str = "recognition"  # or "detection"

    # ---- derived sizes (SURVEY.md section 8 notation) ----
    @property
    def E(self):  # transformer width is 2*d_model (rec tim.py:115-121)
        return 2 * self.d_model

    @property
    def FF(self):
        return self.d_model * self.feedforward_scale

    @property
    def F(self):  # feature tokens per window (rec tim.py:88)
        return 2 * self.num_feats if self.input_modality == "audio_visual" else self.num_feats

    @property
    def has_visual_queries(self):
        if self.input_modality == "audio_visual":
            return "visual" in self.data_modality
        return self.input_modality == "visual"

    @property
    def has_audio_queries(self):
        if self.input_modality == "audio_visual":
            return "audio" in self.data_modality
        return self.input_modality == "audio"

    @property
    def vn(self):
        """Number of visual query groups per visual interval (verb, noun, action)."""
        if self.variant == "detection":
            return 1
        return 3 if self.include_verb_noun else 1

    def num_queries(self, nv, na):
        q = 0
        if self.has_visual_queries and nv > 0:
            q += self.vn * nv
        if self.has_audio_queries and na > 0:
            q += na
        return q


# Named configurations of BASELINE.json / SURVEY.md section 8.
def named_config(name: str) -> TimConfig:
    if name == "C1":
        return TimConfig(d_model=256, nhead=4, num_layers=2, input_modality="visual",
                         data_modality="visual")
    if name in ("C2a", "C5"):
        return TimConfig()
    if name == "C2b":
        return TimConfig(num_feats=75)
    if name == "C3":
        return TimConfig(num_class=[63, 17], include_verb_noun=False, feat_drop=0.1, seq_drop=0.1)
    if name == "C4":
        return TimConfig(num_class=(97, 44), visual_input_dim=2048, include_verb_noun=False,
                         data_modality="visual", variant="detection")
    if name == "tiny":
        return TimConfig(num_class=[[7, 11, 13], 5], visual_input_dim=24, audio_input_dim=40,
                         d_model=32, nhead=2, num_layers=2, num_feats=6)
    raise KeyError(name)
