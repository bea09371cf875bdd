"""bench.py's parts: byte accounting and ceilings (roofline), rocprofv3 counter passes (counters), the line's config / exchange
blocks (record), the CPU baseline and the N = 1 legs (legs), the N > 1 sweep and denominators (dp_legs), the multi-rank launch and
control plane (ranks), the JSON line and its wall-clock budget (emit)."""
