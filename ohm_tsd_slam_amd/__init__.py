"""ohm_tsd_slam_amd -- MI355X (gfx950) implementation of ohm_tsd_slam's per-scan hot path.

The product is ``lib/libtsd_hip.so`` (hand-written HIP kernels behind the C ABI of ``include/tsd_hip.h``)
and the C++ facade in ``csrc/host`` that mirrors ``ThreadLocalize`` / ``ThreadMapping``.  The Python
modules here are thin ctypes drivers for tests and benchmarks (``capi``), the synthetic worlds of the
measurement plan (``synth``) and the multi-GPU occupancy merge (``multigpu``).
"""
from . import capi, synth  # noqa: F401

__all__ = ["capi", "synth"]
