#!/bin/bash
# Whose frame does a rocprofv3-profiled run of config C4 die in at exit (rc 139, after its output files are complete)?
# Runs the kernel-trace pass of tools/profile_bench.sh for C4 with tools/segv_trace.c installed (bench.py loads it when
# FMCMC_SEGV_TRACE=1; the program itself stays directly after `--`), and, for comparison, the same command unprofiled.
# Output: gpurun_out/c4_exit_trace.txt (frames), gpurun_out/c4_exit_trace.log (return codes).
mkdir -p $GRAFT_REPO_ROOT/gpurun_out $GRAFT_REPO_ROOT/tools/exp_bin
[ -f $GRAFT_REPO_ROOT/tools/exp_bin/libsegv_trace.so ] || gcc -shared -fPIC -O1 $GRAFT_REPO_ROOT/tools/segv_trace.c -o $GRAFT_REPO_ROOT/tools/exp_bin/libsegv_trace.so
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c4_exit
rm -rf $OUT; mkdir -p $OUT
rm -f $GRAFT_REPO_ROOT/gpurun_out/c4_exit_trace.txt
cd /tmp && export TMPDIR=/tmp
export FMCMC_SEGV_TRACE=1 FMCMC_SEGV_TRACE_FILE=$GRAFT_REPO_ROOT/gpurun_out/c4_exit_trace.txt
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs"
{
  $CMD > $OUT/plain.stdout 2> $OUT/plain.err; echo "unprofiled rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/kt.stdout 2> $OUT/kt.err; echo "rocprofv3 --kernel-trace --stats rc=$?"
  ls $OUT/kt/*/ 2>/dev/null | head -5
} > $GRAFT_REPO_ROOT/gpurun_out/c4_exit_trace.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/c4_exit_trace.log
touch $GRAFT_REPO_ROOT/gpurun_out/c4_exit_trace.txt
head -60 $GRAFT_REPO_ROOT/gpurun_out/c4_exit_trace.txt
