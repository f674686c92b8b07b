#!/bin/bash
# A/B on one box: builds of the tiled kernel's epilogue (vllm_pq_amd/libpq_<name>.so; base = commit 26202d7,
# epi0..epi2 = -DTL_EPI=0..2, amd = the library), alternating, three rounds.
set -o pipefail
LIBS=${LIBS:-"base epi0 epi1 epi2 amd"}
for r in 1 2 3; do
  for lib in $LIBS; do
    for spec in "gate_up 4096" "o 4096" "gate_up 256" "gate_up 128"; do
      set -- $spec
      echo -n "$lib "; PQ_AMD_LIB=$PWD/vllm_pq_amd/libpq_$lib.so timeout -k 10 120 python tools/kbench.py --shape $1 --tokens $2 --workspace 2>/dev/null | grep -v amdgpu || exit 1
    done
  done
done
