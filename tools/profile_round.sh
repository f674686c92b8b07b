#!/bin/bash
# Round profile on the GPU box: bench line, rocprofv3 kernel trace + stats, three separate PMC passes.
# Usage (from the repo root, on the GPU box):  bash tools/profile_round.sh <tag> [commit]
set -e -o pipefail
TAG=${1:-r01}
export PQ_COMMIT=${2:-${PQ_COMMIT:-}}     # the commit under test (the GPU box has no .git): recorded in traffic.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
echo "[1/6] bench.py"; timeout -k 10 500 python3 "$R/bench.py" > "$O/bench.json"
echo "[2/6] kernel trace"; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -- python3 "$R/bench.py" --no-cpu-baseline --steps 3 --warmup 1 --launch-log "$O/launches.json" > "$O/bench_under_rocprof.json"
SMALL="--no-cpu-baseline --no-mid --layers 2 --steps 1 --warmup 1"   # (the mid-size batches would mix other launches into the per-grid medians)
echo "[3/6] pmc FETCH_SIZE"; timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" $SMALL > /dev/null
echo "[4/6] pmc WRITE_SIZE"; timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" $SMALL > /dev/null
echo "[5/6] pmc SQ"; timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$O/pmc_sq" -- python3 "$R/bench.py" $SMALL > /dev/null
echo "[6/6] summaries"
python3 "$R/tools/proftrace.py" "$O/kt" --launch-log "$O/launches.json" > "$O/kernel_trace_summary.txt"
rm -f "$O/launches.json"
for p in fetch write sq; do echo "== $p"; python3 "$R/tools/pmcsum.py" "$O/pmc_$p"; done > "$O/pmc_summary.txt"
python3 "$R/tools/traffic.py" "$O/pmc_fetch" "$O/pmc_write" "$TAG" > "$O/traffic.json"
find "$O/kt" -name "*kernel_stats.csv" -exec cp {} "$O/kernel_stats.csv" \;
# raw traces are large; keep only the summaries in what travels back
rm -rf "$O/kt" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq"
cat "$O/bench.json"
