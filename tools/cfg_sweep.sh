#!/bin/bash
# every tile shape of the tiled kernel on the model shapes at prefill sizes, against the default choice
set -o pipefail
for spec in "qkv 4096" "o 4096" "down 4096" "gate_up 4096" "qkv 2048" "o 2048" "down 2048" "qkv 1024" "o 1024" "down 1024" "gate_up 1024"; do
  set -- $spec
  echo -n "default "; timeout -k 10 100 python tools/kbench.py --shape $1 --tokens $2 --workspace 2>/dev/null | grep -v amdgpu || exit 1
  for c in 0 1 2 3 6; do
    echo -n "cfg$c    "; PQ_AMD_FORCE_KERNEL=tiled PQ_AMD_TILED_CFG=$c timeout -k 10 100 python tools/kbench.py --shape $1 --tokens $2 --workspace 2>/dev/null | grep -v amdgpu || exit 1
  done
done
