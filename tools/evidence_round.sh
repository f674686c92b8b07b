set -o pipefail
O=gpurun_out/evid_r02; mkdir -p $O
timeout -k 10 120 ./tools/mfma_peak > $O/mfma_peak.txt 2>&1 && echo mfma ok &&
timeout -k 10 200 ./tools/latlab > $O/latlab.txt 2>&1 && echo latlab ok &&
timeout -k 10 200 ./tools/decode_lab > $O/decode_lab.txt 2>&1 && echo dlab ok &&
timeout -k 10 200 ./tools/decode_lab s > $O/decode_stamps.txt 2>&1 && echo dstamps ok &&
timeout -k 10 300 python -u tools/decode_ab.py --out $O/decode_ab.csv > $O/decode_ab.txt 2>&1 && echo ab ok &&
timeout -k 10 900 python -u tools/sweep.py --out $O/sweep.csv > $O/sweep.txt 2>&1 && echo sweep ok
