"""dldkd_amd: MI355X-native (gfx950) implementation of DL-DKD's scoring + distillation hot path.

Drop-in for the reference's `method.model` / `method.eval` surface on that path; the compute is in
libdldkd_hip.so (hand-written HIP, C ABI in include/dldkd_hip.h).  No CPU fallback.
"""
from . import native  # noqa: F401

__all__ = ["native"]
