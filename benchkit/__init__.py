"""bench.py's parts: byte accounting and ceilings (roofline), rocprofv3 counter passes (counters), the optional legs and the
CPU baseline (legs), the multi-rank launch and control plane (ranks), the JSON line and its wall-clock budget (emit)."""
