#!/bin/bash
# scripts/kernel_regs.sh <file.hip> [filter]: VGPRs / spill bytes / occupancy / LDS of every kernel in a translation unit
cd "$(dirname "$0")/../pytorch-glow_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 |
  awk '/Function Name:/ {name=$5} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /ScratchSize/ {s=$5} /Occupancy/ {o=$5} /LDS Size/ {print name, "vgpr", v, "agpr", a, "scratch", s, "occ", o, "lds", $6}' |
  while read n rest; do echo "$(echo $n | c++filt | cut -c1-100) $rest"; done | grep -i "${2:-.}"
