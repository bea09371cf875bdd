#!/bin/bash
# VGPRs / occupancy / LDS / scratch of every kernel in a csrc file: kernel_resources.sh [grep-pattern] [file.hip, default fm_forward.hip]
cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC --cuda-device-only -c /root/repo/sparkfm_amd/csrc/${2:-fm_forward.hip} \
  -I/root/repo/include -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "error|Function Name|VGPRs:|Occupancy|LDS Size|ScratchSize" | paste - - - - - |
  sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s#/root/repo/sparkfm_amd/csrc/[a-z_]*.hip:[0-9]*:[0-9]*: remark:##g; s/Function Name: //' |
  c++filt | sed 's/fmhip::(anonymous namespace):://; s/(fmhip::[A-Za-z]*)//; s/void //; s/\[bytes\/[a-z]*\]//g; s/\[waves\/SIMD\]//' | tr -s ' \t' ' ' | grep -E "${1:-.}"
