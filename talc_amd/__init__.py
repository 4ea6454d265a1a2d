"""talc_amd — MI355X-native implementation of TALC's per-long-read correction hot path.

The product is the C-ABI library talc_amd/_build/libtalc_hip.so (include/talc_hip.h) plus the
drop-in `talc` CLI; this package is the thin ctypes binding used by tests and bench.py.
"""
__all__ = ["build", "synth"]
