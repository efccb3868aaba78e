#!/usr/bin/env python3
"""Register / scratch / spill / occupancy table of the engine's kernels, from hipcc -Rpass-analysis=kernel-resource-usage.

  python tools/resource_table.py [--tag r03] [--all]      (runs here: hipcc cross-compiles gfx950 without a GPU)

Writes profiles/<tag>_resources.json (every kernel) and profiles/<tag>_resources.md (the product kernels of the four GPU
configs first, then every kernel that uses scratch).  The compile is the product's own command line (fmcmc_amd/build.py
FLAGS) plus `--cuda-device-only -c` and the remark flag, so the numbers are those of the shipped code object."""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fmcmc_amd import build as B  # noqa: E402

FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
          "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
          "LDS Size [bytes/block]": "static_lds_bytes"}
# the kernels the four GPU configs of BASELINE.md section 4 run on (bench.py fails if the dispatcher picks another one)
PRODUCT = [("C2 headline", r"mh_sweep_mfma<1, 1, 20, false>"), ("C3", r"mh_sweep_spec<3, 20, 3>"),
           ("C3' kernel_ram k=5", r"mh_sweep_spec<3, 20, 4>"), ("C4", r"mh_sweep_wide2<4, 3>"),
           ("C4 slice product", r"shard_columns_mfma<2, 3, 12>"),
           ("C5", r"mh_sweep_kernel<4, -1, 0, 2, 2, 1>"), ("rng stream", r"rng_fill_kernel"),
           ("Gelman window covariance", r"gelman_chain_mfma"), ("Gelman chain sum", r"gelman_sum_kernel")]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(anonymous namespace\)::", "", s).split("(")[0].replace("void ", "") for s in out.splitlines()]


def collect(extra_flags=()):
    rows = []
    t0 = time.time()
    for src in B.SRC:
        cmd = [B.HIPCC] + [f for f in B.FLAGS if f != "-shared"] + list(extra_flags) + \
              ["--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"mangled": m.group(1), "source": os.path.basename(src)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+) \[-Rpass", line)
            if m and cur is not None and m.group(1) in FIELDS:
                cur[FIELDS[m.group(1)]] = int(m.group(2)) if m.group(2).isdigit() else m.group(2)
    for r, d in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = d
    return rows, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--flags", default="", help="extra compile flags (e.g. -DFMCMC_AB)")
    args = ap.parse_args()
    rows, secs = collect(args.flags.split())
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    rec = {"head": head, "flags": B.FLAGS + args.flags.split(), "device_compile_seconds": round(secs, 1), "kernels": rows,
           "n_kernels": len(rows), "n_with_scratch": sum(1 for r in rows if r.get("scratch_bytes_per_lane", 0))}
    json.dump(rec, open(os.path.join(ROOT, "profiles", "%s_resources.json" % args.tag), "w"), indent=1)
    cols = ["vgprs", "agprs", "sgprs", "scratch_bytes_per_lane", "vgpr_spills", "sgpr_spills", "occupancy_waves_per_simd"]
    hdr = "| role | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | SGPR spills | waves/SIMD |\n|---|---|---|---|---|---|---|---|---|\n"
    md = ["# Kernel resources (%s, commit %s)\n" % (args.tag, head),
          "`python tools/resource_table.py --tag %s`: hipcc `-Rpass-analysis=kernel-resource-usage` on the product's own "
          "command line; %d kernels / device functions in the code object, %d of them with scratch; device compile %.0f s.\n"
          % (args.tag, len(rows), rec["n_with_scratch"], secs),
          "## Product kernels of the GPU configs\n", hdr]
    for role, pat in PRODUCT:
        for r in rows:
            if pat in r["name"]:
                md.append("| %s | `%s` | %s |\n" % (role, r["name"], " | ".join(str(r.get(c, "")) for c in cols)))
    md.append("\n## Every kernel with scratch\n\n" + hdr)
    for r in sorted(rows, key=lambda r: -int(r.get("scratch_bytes_per_lane", 0) or 0)):
        if r.get("scratch_bytes_per_lane", 0):
            md.append("| | `%s` | %s |\n" % (r["name"], " | ".join(str(r.get(c, "")) for c in cols)))
    open(os.path.join(ROOT, "profiles", "%s_resources.md" % args.tag), "w").write("".join(md))
    print("".join(md[:4 + len(PRODUCT) + 2]))


if __name__ == "__main__":
    main()
