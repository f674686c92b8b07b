set -e
run() { echo "== CFG=$1 $2 T=$3"; PQ_AMD_FORCE_KERNEL=stream PQ_AMD_STREAM_CFG=$1 timeout -k 10 100 python tools/kbench.py --shape $2 --tokens $3 --workspace 2>&1 | grep -v amdgpu.ids; }
run 1x1 o,qkv 32,64
run 2x1 o,qkv 32,64,128
run 4x1 o,qkv,down 32,64,128
run 8x1 o,qkv,gate_up,down 32,64,128
run 7x1 gate_up 32,64
