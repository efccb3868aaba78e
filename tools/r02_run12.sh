set -o pipefail
cd /root/repo
bash tools/profile_bench.sh r02_b_c2 c2 full > gpurun_out/prof_r02_b_c2.log 2>&1; tail -5 gpurun_out/prof_r02_b_c2.log
bash tools/profile_bench.sh r02_b_c3 c3 stats > gpurun_out/prof_r02_b_c3.log 2>&1
bash tools/profile_bench.sh r02_b_c5 c5 stats > gpurun_out/prof_r02_b_c5.log 2>&1
bash tools/profile_bench.sh r02_b_c4 c4 full > gpurun_out/prof_r02_b_c4.log 2>&1; tail -5 gpurun_out/prof_r02_b_c4.log
for c in c2 c3 c4 c5; do python3 -c "
import json; d=json.load(open('gpurun_out/prof_r02_b_$c/summary.json')); print('$c', [(r['Name'][:60], r['Calls'], r['AverageNs']) for r in d.get('kernel_stats', [])[:4]]); print('   hbm', d.get('hbm_bytes_per_launch'), {k: v for k, v in d.get('pmc_per_launch_mean', {}).get('mh_sweep', {}).items() if 'MFMA' in k or 'SIZE' in k or 'BUSY_CU' in k})"; done
