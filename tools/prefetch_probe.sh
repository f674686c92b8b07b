cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in none right wrong early; do
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pf_$m -- python3 $R/tools/prefetch_probe.py $m 16 > $R/gpurun_out/pf_$m.log 2>&1 || exit 1
done
