set -e
mkdir -p gpurun_out/r03
for b in w32 w64 w32 w64; do timeout -k 10 120 tools/exp_bin/exp_mfma_$b 10000 3; done 2>&1 | tee gpurun_out/r03/exp_mfma_w64.txt
