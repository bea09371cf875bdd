#!/bin/bash
# Forward time against the order rows are walked in: stored order (default at Kp = 32), the whole batch sorted by length,
# or sorted inside windows of W rows (the eight rows a wave walks together then have equal length, and stay neighbours;
# windows alternate longest-first / shortest-first so that every workgroup sees both ends).
set -e
for cfg in C3 C2; do
  echo "== $cfg stored order"; timeout -k 10 200 python3 tools/fwd_modes_time.py $cfg
  for w in 64 256 1024; do
    echo "== $cfg FMHIP_ORDER_ALL=1 window $w"; FMHIP_ORDER_ALL=1 FMHIP_ORDER_WINDOW=$w timeout -k 10 200 python3 tools/fwd_modes_time.py $cfg
  done
done
echo "== C5 window sweep (Kp = 64: sorted by default)"
for w in 0 64 256 1024; do FMHIP_ORDER_WINDOW=$w timeout -k 10 200 python3 tools/fwd_modes_time.py C5; done
