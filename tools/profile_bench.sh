#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the contract bench command for one config.
#   bash tools/profile_bench.sh <tag> [config=c2] [full|stats]
# Outputs compact summaries under gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/.
# (rocprofv3 is given the program itself after `--`: no env / bash -c hop, see the pool's rules.)
set -u
TAG=${1:-r02}
CFG=${2:-c2}
MODE=${3:-full}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $CMD > $OUT/bench_under_trace.json 2> $OUT/kt.err
if [ "$MODE" = "full" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_misc -- $CMD > /dev/null 2> $OUT/pmc_misc.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_mfma -- $CMD > /dev/null 2> $OUT/pmc_mfma.err
fi
cd $GRAFT_REPO_ROOT
python3 - "$OUT" "$CFG" <<'PY'
import csv, glob, json, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
summary = {"config": cfg}
ks = glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    summary["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:8]]
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        key = "mh_sweep" if "mh_sweep" in name else ("rng_fill" if "rng_fill" in name else ("gelman_chain" if "gelman_chain" in name else None))
        if key:
            pm[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary["pmc_per_launch_mean"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pm.items()}
summary["pmc_launches"] = {k: {c: len(v) for c, v in d.items()} for k, d in pm.items()}
sw = summary["pmc_per_launch_mean"].get("mh_sweep", {})
if "FETCH_SIZE" in sw and "WRITE_SIZE" in sw:
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KB) reports half the bytes of wide streaming reads on gfx950 -> x2; WRITE_SIZE (KB) exact
    summary["hbm_bytes_per_launch"] = sw["FETCH_SIZE"] * 1024 * 2 + sw["WRITE_SIZE"] * 1024
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:3500])
PY
