"""zbra_amd — Zbra's integer column codec (frame of reference, zig-zag, bit-packing) on MI355X.

`zbra_amd.compression` mirrors the reference's `zbra_core::compression` module over host arrays;
`zbra_amd.device` drives the device-resident entry points on torch tensors (torch is used for
device memory and streams only).  Both call libzbra_hip.so; neither has a CPU fallback.
"""
from . import _lib  # noqa: F401  (raises if the HIP library is missing)
from . import compression  # noqa: F401

__all__ = ["compression"]
