set -e
for m in 0 4096 8192 12288 16384 24576; do echo "== mode=$m"; FMCMC_AMD_DEBUG=mode=$m timeout -k 10 200 python tools/bench_c4.py 1000 2>&1 | grep kernel_ram; done
