#!/bin/bash
# Times one sweep shape under every forced tiled-kernel tile shape: tools/cfg_ab.sh <shape substring> <token counts>
for c in 0 1 2 3 4; do
  echo "== PQ_AMD_TILED_CFG=$c"
  PQ_AMD_FORCE_KERNEL=tiled PQ_AMD_TILED_CFG=$c timeout -k 10 120 python tools/sweep.py --only "$1" --ts "$2" --out gpurun_out/cfg_ab_$c.csv || exit 1
done
