"""dtlr_amd -- MI355X-native (gfx950) inference engine for detection-based text-line recognition
(the hot path of raphael-baena/DTLR): hand-written HIP kernels behind a C ABI (libdtlr_hip.so,
include/dtlr_hip.h) exposed under the reference's own operator/module names."""
from __future__ import annotations

import sys

from .config import DTLRConfig  # noqa: F401

__all__ = ["DTLRConfig", "install_dropin"]


def install_dropin() -> None:
    """Register this package's operator module under the name the reference imports
    (`import MultiScaleDeformableAttention as MSDA`, ops/functions/ms_deform_attn_func.py:18), so a
    reference checkout runs its own MSDeformAttnFunction on the HIP kernel unchanged."""
    from . import MultiScaleDeformableAttention as m
    sys.modules["MultiScaleDeformableAttention"] = m
