set -e
mkdir -p gpurun_out/r03
for b in late_mb12 late_mb12_t2 late_mb16_t2 mb8 late_mb12 late_mb12_t2 late_mb16_t2; do timeout -k 10 120 tools/exp_bin/exp_mfma_$b 10000 3; done 2>&1 | tee gpurun_out/r03/exp_mfma_trims2.txt
