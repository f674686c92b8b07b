#!/bin/bash
# Evidence pass on the GPU box: launch-path micro-benchmarks, decode stamps / ablations / A-B, tiled stamps, bare-MFMA
# ceilings, operand-statistics probe, roofline sweep, exhaustive search.  Usage: bash tools/evidence_round.sh <tag>
set -o pipefail
TAG=${1:-r02}
O=gpurun_out/evid_$TAG; mkdir -p $O
timeout -k 10 120 ./tools/mfma_peak > $O/mfma_peak.txt 2>&1 && echo mfma ok &&
timeout -k 10 120 ./tools/bin/mfma_power > $O/mfma_issue_order.txt 2>&1 && echo order ok &&
timeout -k 10 200 ./tools/latlab > $O/latlab.txt 2>&1 && echo latlab ok &&
timeout -k 10 100 ./tools/bin/ingest_lab > $O/ingest_lab.txt 2>&1 && echo ingest ok &&
timeout -k 10 200 ./tools/decode_lab > $O/decode_lab.txt 2>&1 && echo dlab ok &&
timeout -k 10 200 ./tools/decode_lab s > $O/decode_stamps.txt 2>&1 && echo dstamps ok &&
timeout -k 10 120 ./tools/bin/tiled_lab 4096 s > $O/tiled_stamps.txt 2>&1 && echo tstamps ok &&
timeout -k 10 300 python -u tools/powerprobe.py 2>&1 | grep -v amdgpu.ids > $O/powerprobe.txt && echo power ok &&
timeout -k 10 300 python -u tools/decode_ab.py --out $O/decode_ab.csv > $O/decode_ab.txt 2>&1 && echo ab ok &&
timeout -k 10 600 python -u tools/sweep.py --out $O/sweep.csv > $O/sweep.txt 2>&1 && echo sweep ok &&
timeout -k 10 550 python -u tools/autotune.py --full --out $O/autotune.csv > $O/autotune.txt 2>&1 && echo autotune ok &&
timeout -k 10 550 python -u tools/autotune.py --square --out $O/autotune_square.csv > $O/autotune_square.txt 2>&1 && echo square ok
