set -o pipefail
cd /root/repo
python tools/stamp_mfma.py > gpurun_out/r02_stamp_mfma.log 2>&1; grep -v amdgpu.ids gpurun_out/r02_stamp_mfma.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "full_size or resident or streamed_equals" > gpurun_out/r02_gpu_tests_h.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02_gpu_tests_h.log
