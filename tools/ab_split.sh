for r in 1 2 3; do
  for s in 0 auto; do
    for T in 2048 2560 3072; do
      if [ $s = 0 ]; then export PQ_AMD_TILED_SPLIT=0; else unset PQ_AMD_TILED_SPLIT; fi
      echo -n "split=$s "; timeout -k 10 120 python tools/kbench.py --shape gate_up --tokens $T --workspace 2>/dev/null | grep -v amdgpu || exit 1
    done
  done
done
