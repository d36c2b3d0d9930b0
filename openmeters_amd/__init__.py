"""openmeters_amd — MI355X-native DSP hot path behind OpenMeters' processor API.

The product is the C-ABI shared library ``openmeters_amd/csrc/libomx_hip.so`` (hand-written HIP
for gfx950, declared in ``include/omx.h``).  This package is only the Python-side mirror of the
reference's processor interface on top of that library (``capi.py``) plus the multi-GPU sharding
helper (``sharding.py``).  There is no CPU fallback: if the HIP library is missing the import of
``api()`` raises, and every entry point returns ``OMX_ERR_NO_DEVICE`` without a gfx950 device.
"""
from __future__ import annotations

import os

from . import capi
from .capi import (  # noqa: F401  (re-exported: the reference's config / snapshot vocabulary)
    AudioBlock, LoudnessConfig, OscilloscopeConfig, SpectrogramConfig, SpectrumConfig, StereometerConfig, WaveformConfig,
    OmxError,
)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMX_HIP_LIB") or os.path.join(_HERE, "csrc", "libomx_hip.so")  # OMX_HIP_LIB: A/B builds (tuning)
_API = None


def api() -> capi.Api:
    """The product C-ABI (prefix ``omx_``).  Fails loudly when the HIP extension is not built."""
    global _API
    if _API is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  openmeters_amd has no CPU fallback.")
        # PyTorch wheels bundle their own libamdhip64; a process must not end up with two HIP runtimes (the second one
        # sees no device).  Loading torch first makes libomx_hip.so's NEEDED libamdhip64.so.7 resolve to the copy that is
        # already mapped, whatever order the caller imports things in.  Hosts without torch (the Rust binding) have one
        # runtime anyway.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _API = capi.Api(LIB_PATH, "omx_")
    return _API


def device_available() -> bool:
    import ctypes
    return bool(api().fn("device_available", ctypes.c_int, [])())
