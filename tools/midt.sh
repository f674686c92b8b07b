set -e
run() { echo "== $1 T=$2"; timeout -k 10 100 python tools/kbench.py --shape $1 --tokens $2 --workspace 2>&1 | grep -v amdgpu.ids; }
run gate_up,qkv,o,down 17,24,32,48,64
