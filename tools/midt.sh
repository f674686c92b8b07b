set -e
run() { echo "== CFG=$1 $2 T=$3"; PQ_AMD_FORCE_KERNEL=$4 PQ_AMD_STREAM_CFG=$1 timeout -k 10 100 python tools/kbench.py --shape $2 --tokens $3 --workspace 2>&1 | grep -v amdgpu.ids; }
run 4x4 o,qkv,down 128,192,256,384,512 stream
run 2x4 o,qkv,down 128,256,512 stream
run "" o,qkv,down 128,192,256,384,512 tiled
