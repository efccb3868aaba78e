#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of the contract bench command for one config.
#   bash tools/profile_bench.sh <tag> [config=c2] [full|stats]
# Outputs under gpurun_out/prof_<tag>/ : kernel_stats.csv (the --stats table of THE pass this call ran), summary.json,
# latest_pmc_<config>.json (full mode: what bench.py reports as roofline.traffic, with the commit it measured).
# Copy what should be judged into profiles/ (tools/collect_profiles.sh does).
# Every rocprofv3 pass writes into a directory of its own that is emptied first: rocprofv3 names its files by pid, and a
# directory that survives from an earlier call would hand the summary step a stale pass.
# (rocprofv3 is given the program itself after `--`: no env / bash -c hop, see the pool's rules.)
set -u
TAG=${1:-r03}
CFG=${2:-c2}
MODE=${3:-full}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-default-path"
pass() {   # pass <name> <rocprofv3 options...>
  local name=$1; shift
  rm -rf $OUT/$name
  rocprofv3 "$@" --output-format csv -d $OUT/$name -- $CMD > $OUT/$name.stdout 2> $OUT/$name.err
  echo "pass $name rc=$?" >> $OUT/passes.log
}
pass kt --kernel-trace --stats
if [ "$MODE" = "full" ]; then
pass pmc_fetch --pmc FETCH_SIZE
pass pmc_write --pmc WRITE_SIZE
pass pmc_sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
pass pmc_sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES
pass pmc_sq3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES
pass pmc_misc --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
fi
cd $GRAFT_REPO_ROOT
python3 - "$OUT" "$CFG" "$TAG" <<'PY'
import csv, glob, json, os, shutil, subprocess, sys, collections
out, cfg, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.environ.get("GRAFT_REPO_ROOT", ".")
head = None
try:
    head = open(os.path.join(root, ".head_at_push")).read().strip()     # written by tools/gpu.sh before the gpurun call (no .git on the box)
except OSError:
    pass
summary = {"config": cfg, "tag": tag, "head": head,
           "command": "bench.py --config %s --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-default-path" % cfg}
ks = sorted(glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True))
assert len(ks) <= 1, "more than one kernel_stats.csv in a fresh pass directory: %s" % ks
sweep_name, sweep_avg = None, None
if ks:
    shutil.copy(ks[0], out + "/kernel_stats.csv")
    rows = list(csv.DictReader(open(ks[0])))
    summary["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:8]]
    for r in rows:
        if "mh_sweep" in r["Name"]:
            sweep_name, sweep_avg = r["Name"], float(r["AverageNs"])
            break
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
d = {}
if "FETCH_SIZE" in sw and "WRITE_SIZE" in sw:
    # MI355X_MICROARCH.md, HBM: FETCH_SIZE (KB) reports half the bytes of wide streaming reads on gfx950 -> x2; WRITE_SIZE (KB) exact
    summary["hbm_bytes_per_launch"] = sw["FETCH_SIZE"] * 1024 * 2 + sw["WRITE_SIZE"] * 1024
# derived figures a reader would otherwise recompute (SQ_* cycle counters are in quad-cycles per MI355X_MICROARCH.md;
# SQ_VALU_MFMA_BUSY_CYCLES in cycles; every ratio below is between counters of ONE pass)
if sw.get("SQ_WAVE_CYCLES"):
    for c in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if c in sw: d[c + "/SQ_WAVE_CYCLES"] = sw[c] / sw["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in sw and sw.get("SQ_BUSY_CYCLES"):
        d["valu_insts_per_wave_quadcycle"] = sw["SQ_INSTS_VALU"] / sw["SQ_WAVE_CYCLES"]
if sw.get("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_share_of_lds_cycles"] = sw.get("SQ_LDS_BANK_CONFLICT", 0.0) / sw["SQ_LDS_IDX_ACTIVE"]
if sw.get("SQ_BUSY_CU_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in sw:
    d["mfma_busy_share_of_simd_cycles"] = sw["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * sw["SQ_BUSY_CU_CYCLES"])
summary["derived"] = d
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
if "hbm_bytes_per_launch" in summary and sweep_name:
    json.dump({"hbm_bytes_per_launch": summary["hbm_bytes_per_launch"], "kernel": sweep_name, "kernel_avg_ns_rocprof": sweep_avg,
               "launches": int(min(summary["pmc_launches"]["mh_sweep"].values())), "head": head, "tag": tag,
               "source": "gpurun_out/prof_%s/summary.json -> profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `%s`; "
                         "FETCH_SIZE x2 and KB = 1024 B as MI355X_MICROARCH.md prescribes for gfx950; mean over the sweep kernel's launches)" % (tag, tag, summary["command"])},
              open(out + "/latest_pmc_%s.json" % cfg, "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("config", "tag", "head", "derived") if k in summary}, indent=1))
print(json.dumps(summary.get("kernel_stats", [])[:2], indent=1)[:1500])
print(json.dumps(sw, indent=1)[:2500])
PY
