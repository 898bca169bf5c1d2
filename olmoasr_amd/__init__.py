"""olmoasr_amd -- MI355X-native (gfx950 HIP) implementation of the OLMoASR training hot path.

Mirrors the reference's Python surface for this path (olmoasr/__init__.py:17-21, olmoasr/model.py); all
arithmetic runs in liboasr.so (olmoasr_amd/csrc, C ABI in include/oasr.h).  There is no CPU fallback."""
__version__ = "0.1.0"
