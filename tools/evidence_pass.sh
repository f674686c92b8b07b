#!/bin/bash
# Short evidence pass on the GPU box (needs the lab binaries under tools/bin, which .gpurunignore keeps out of the push by default): tiled stamps, mid-batch stamps, roofline sweep, exhaustive search, bench line.
# Usage: bash tools/evidence_pass.sh <tag>
set -o pipefail
TAG=${1:-pass}
O=gpurun_out/evid_$TAG; mkdir -p $O
timeout -k 10 120 ./tools/bin/tiled_lab 4096 s > $O/tiled_stamps.txt 2>&1 && echo tstamps ok &&
timeout -k 10 200 ./tools/bin/wide_lab > $O/mid_stamps.txt 2>&1 && echo mid ok &&
timeout -k 10 700 python -u tools/sweep.py --out $O/sweep.csv > $O/sweep.txt 2>&1 && echo sweep ok &&
timeout -k 10 550 python -u tools/autotune.py --full --out $O/autotune.csv > $O/autotune.txt 2>&1 && echo autotune ok &&
timeout -k 10 300 python -u bench.py > $O/bench.json 2> $O/bench.err && echo bench ok
