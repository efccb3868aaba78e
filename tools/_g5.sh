set -e
mkdir -p gpurun_out/r03
for b in spec_base spec_prio1 spec_prio3 spec_prio3_stamp spec_base_ram spec_prio3_ram; do timeout -k 10 120 tools/exp_bin/exp_$b 10000 2; done 2>&1 | tee gpurun_out/r03/exp_spec_prio.txt
