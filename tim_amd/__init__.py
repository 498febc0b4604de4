"""tim_amd — MI355X (gfx950) implementation of the TIM encoder hot path.

`tim_amd.TIM` / `tim_amd.build_model` mirror `time_interval_machine.models.{tim,build}` of the
reference; the arithmetic lives in `libtimhip.so` (tim_amd/csrc, C ABI in include/timhip.h).
"""
from .config import TimConfig, named_config  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require torch on a build box
    if name == "TIM":
        from .tim import TIM
        return TIM
    if name == "build_model":
        from .build import build_model
        return build_model
    if name == "DetectionTIM":
        from .detection import TIM
        return TIM
    raise AttributeError(name)
