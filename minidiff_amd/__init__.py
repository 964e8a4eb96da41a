"""minidiff_amd — MI355X (gfx950) backend for the minidiff autodiff engine.

See DESIGN.md. Nothing is loaded at import; the first array operation binds
the process to ``minidiff_amd/libmdhip.so`` and raises ImportError if the HIP
extension has not been built (there is no CPU fallback).
"""
__all__ = ["ndarray", "hip_backend", "tape", "_capi"]
