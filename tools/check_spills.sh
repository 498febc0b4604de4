#!/bin/bash
# Which kernels of tim_amd/csrc spill registers?  (hipcc cross-compiles: runs in the build container, no GPU.)
#   bash tools/check_spills.sh [file.hip ...]          default: every .hip of the library
# The one-block-per-CU GEMM kernels run three waves per SIMD at 168 VGPRs: their epilogues sit at that limit, and an innocent
# change (a loop around the kernel body, a value kept live across the epilogue) has made the compiler spill 20-132 registers
# there without any warning - DESIGN.md section 5d.  Run this after every change to gemm_pp.hip / wgrad_pp.hip.
cd "$(dirname "$0")/../tim_amd/csrc" || exit 1
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(*.hip)
for f in "${files[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /tmp/check_spills.o -Rpass-analysis=kernel-resource-usage 2> /tmp/check_spills.txt
  n=$(grep -c 'VGPRs Spill' /tmp/check_spills.txt); s=$(grep -c 'VGPRs Spill: [1-9]' /tmp/check_spills.txt)
  echo "$f: $s of $n kernels spill VGPRs"
  grep -B8 'VGPRs Spill: [1-9]' /tmp/check_spills.txt | grep -E 'Function Name|VGPRs Spill' | sed 's/.*remark: *//;s/\[-Rpass.*//' | paste - - | c++filt | cut -c1-200
done
