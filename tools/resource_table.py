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
PRODUCT = [("C2 headline", r"mh_sweep_mfma<1, 1, 20, false, false, false>"),
           ("C2 shape, 256 chains or fewer per GPU (latency form)", r"mh_sweep_lat<1, 3, 20>"), ("C2 shape, > 4 GiB of samples", r"mh_sweep_mfma<1, 1, 20, false, true, false>"),
           ("C3", r"mh_sweep_spec<3, 20, 3, 1, false>"),
           ("C3' kernel_ram k=5", r"mh_sweep_spec<3, 20, 4, 1, false>"), ("C4", r"mh_sweep_wide2<4, 3>"),
           
           ("C5 (observation-sharded, owners in the hand-overs' shadow)", r"mh_sweep_logit2<2>"),
           ("C5 shape on the general kernel's observation-sharded form (stream-fed: knob shadow=0, kernel_adapt / kernel_ram)", r"mh_sweep_kernel<4, -1, 2, 2, 2, 1, true>"),
           ("the same with in-kernel variates (streams beyond 1 GiB)", r"mh_sweep_kernel<4, -1, 2, 2, 2, 1, false>"),
           ("C5 shape, chain-sharded form", r"mh_sweep_kernel<4, -1, 0, 2, 2, 1, false>"),
           ("n > 10240, normal kernels", r"mh_sweep_mfma<1, 1, 16, false, false, true>"), ("8 <= p <= 11, normal kernels", r"mh_sweep_mfma<1, 3, 4, false, false, true>"),
           ("12 <= p <= 15, normal kernels", r"mh_sweep_mfma<1, 4, 2, false, false, true>"),
           ("n > 10240, kernel_adapt k = 5", r"mh_sweep_mfma_ad<3, 1, 5, false>"), ("bounded kernel_ram k = 5", r"mh_sweep_mfma_ad<4, 1, 5, true>"),
           ("kernel_adapt, 8 <= p <= 11 (owners' matrices in LDS)", r"mh_sweep_mfma_ad<3, 3, -1, false>"),
           ("64 < k <= 128", r"mh_sweep_bigk"), ("rng stream", r"rng_fill_kernel"),
           ("Gelman window covariance", r"gelman_chain_mfma"), ("Gelman chain sum", r"gelman_sum_kernel")]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"\(anonymous namespace\)::", "", s).split("(")[0].replace("void ", "") for s in out.splitlines()]


def _one_unit(src, extra_flags):
    """remarks and ISA of one translation unit (the product's own flags + --cuda-device-only)"""
    base = [B.HIPCC] + list(B.FLAGS) + list(extra_flags) + ["--cuda-device-only"]
    t0 = time.time()
    err = subprocess.run(base + ["-c", "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"], capture_output=True, text=True).stderr
    secs = time.time() - t0
    asm = "/tmp/fmcmc_amd_%s.s" % os.path.splitext(os.path.basename(src))[0]
    subprocess.run(base + ["-S", src, "-o", asm], capture_output=True, text=True)
    return src, err, secs, asm


_UNITS = None


def units(extra_flags=()):
    global _UNITS
    if _UNITS is None:
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.time()
        with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
            res = list(ex.map(lambda s_: _one_unit(s_, extra_flags), B.sources()))
        _UNITS = (res, time.time() - t0)
    return _UNITS


def collect(extra_flags=()):
    rows = []
    res, wall = units(extra_flags)
    for src, err, secs, _asm in res:
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
    return rows, {"wall_8_jobs": round(wall, 1), "sum_of_units": round(sum(u[2] for u in res), 1),
                  "per_unit": {os.path.basename(u[0]): round(u[2], 1) for u in res}}


# real (noinline) device functions that carry the hot loops of a product kernel: the remark pass reports kernels only, so
# their registers and spill code are read off the ISA (-S): highest VGPR named, scratch instructions, and v_readlane /
# v_writelane (SGPR spill traffic) inside their innermost hot loop (the largest backward-branch body below 400 instructions)
DEVICE_FUNCS = [("C5 observation loop (observation-sharded)", "logit_shard<5, 2, 2>"), ("logistic p = 12 (one observation per pass)", "logit_shard<12, 2, 1>"), ("C5 shape, chain-sharded observation loop", "logit_partials<4, 5>"),
                ("C4 slice product (form T10)", "shard_columns_mfma<2, 3, 12, true>"),
                ("C4 factor update + proposal", "w2_ram_update_propose")]


def isa_functions(extra_flags=()):
    out = []
    funcs, cur = {}, None
    for _src, _err, _secs, asm in units(extra_flags)[0]:
        cur = None
        for line in open(asm):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur = m.group(1)
                if cur in funcs:          # (the same noinline helper in several translation units: the first one stands for all)
                    cur = None
                else:
                    funcs[cur] = []
            elif cur is not None:
                funcs[cur].append(line.rstrip("\n"))
                if line.startswith(".Lfunc_end"):
                    cur = None
    names = list(funcs)
    for role, pat in DEVICE_FUNCS:
        for mangled, dem in zip(names, demangle(names)):
            if dem != pat:
                continue
            body = funcs[mangled]
            ins = [x for x in body if x.startswith("\t") and not x.strip().startswith((";", "."))]
            vg = [int(g) for x in ins for g in re.findall(r"\bv(\d+)\b", x)] + [int(b) for x in ins for _, b in re.findall(r"v\[(\d+):(\d+)\]", x)]
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            best = None
            for i, l in enumerate(body):
                m = re.search(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", l)
                if m and m.group(1) in labels and labels[m.group(1)] < i:
                    loop = [x for x in body[labels[m.group(1)]:i + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
                    if len(loop) < 400 and (best is None or len(loop) > len(best)):
                        best = loop
            cnt = lambda seq, pat_: sum(1 for x in seq if re.search(pat_, x))
            out.append({"role": role, "name": dem, "highest_vgpr": max(vg) if vg else -1, "instructions": len(ins),
                        "scratch_instructions": cnt(ins, "scratch_"), "hot_loop_instructions": len(best or []),
                        "hot_loop_fp64": cnt(best or [], "_f64"), "hot_loop_mfma": cnt(best or [], "v_mfma"),
                        "hot_loop_scratch": cnt(best or [], "scratch_"), "hot_loop_readlane_writelane": cnt(best or [], "v_readlane|v_writelane")})
    # where a product kernel's scratch IS: the remark pass reports a kernel's worst-case stack over everything it can call; what
    # matters is which scratch instructions sit on the path a config executes -- the kernel body (by enclosing loop) and each callee
    calls = []
    dem = dict(zip(names, demangle(names)))
    for role, pat in PRODUCT:
        for mangled in names:
            if pat not in dem[mangled]:
                continue
            body = funcs[mangled]
            ins = [(i, x) for i, x in enumerate(body) if x.startswith("\t") and not x.strip().startswith((";", "."))]
            if not any("s_endpgm" in x for _, x in ins):
                continue                     # (device functions are listed as callees)
            labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            loops = []
            for i, l in enumerate(body):
                m = re.search(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", l)
                if m and m.group(1) in labels and labels[m.group(1)] < i:
                    loops.append((labels[m.group(1)], i))
            sc = [i for i, x in ins if "scratch_" in x]
            in_loop = sum(1 for i in sc if any(a <= i <= b for a, b in loops))
            cal = []
            for l in body:
                m = re.search(r"(_Z\w+)@(?:rel32|gotpcrel32)", l)
                if m and m.group(1) in funcs and m.group(1) not in [c[0] for c in cal]:
                    cb = [x for x in funcs[m.group(1)] if x.startswith("\t") and not x.strip().startswith((";", "."))]
                    if cb:
                        cal.append((m.group(1), dem[m.group(1)], len(cb), sum(1 for x in cb if "scratch_" in x)))
            calls.append({"role": role, "kernel": dem[mangled], "instructions": len(ins), "scratch_instructions": len(sc),
                          "scratch_instructions_inside_a_loop": in_loop,
                          "callees": [{"name": c[1], "instructions": c[2], "scratch_instructions": c[3]} for c in cal]})
    return out, calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r03")
    ap.add_argument("--flags", default="", help="extra compile flags (e.g. -DFMCMC_AB)")
    args = ap.parse_args()
    rows, secs = collect(args.flags.split())
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    fns, calls = isa_functions(args.flags.split())
    rec = {"head": head, "flags": B.FLAGS + args.flags.split(), "device_compile_seconds": secs, "kernels": rows, "device_functions": fns, "scratch_by_path": calls,
           "n_kernels": len(rows), "n_with_scratch": sum(1 for r in rows if r.get("scratch_bytes_per_lane", 0))}
    json.dump(rec, open(os.path.join(ROOT, "profiles", "%s_resources.json" % args.tag), "w"), indent=1)
    cols = ["vgprs", "agprs", "sgprs", "scratch_bytes_per_lane", "vgpr_spills", "sgpr_spills", "occupancy_waves_per_simd"]
    hdr = "| role | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | SGPR spills | waves/SIMD |\n|---|---|---|---|---|---|---|---|---|\n"
    md = ["# Kernel resources (%s, commit %s)\n" % (args.tag, head),
          "`python tools/resource_table.py --tag %s`: hipcc `-Rpass-analysis=kernel-resource-usage` on the product's own "
          "command line, one pass per translation unit; %d kernels in the code objects, %d of them with scratch.  Device compile (remark pass, "
          "8 jobs): %.0f s wall, %.0f s summed over the units (%s); `python -m fmcmc_amd.build` (full compile + link, 8 jobs) takes about a minute.\n"
          % (args.tag, len(rows), rec["n_with_scratch"], secs["wall_8_jobs"], secs["sum_of_units"], ", ".join("%s %.0f" % kv for kv in sorted(secs["per_unit"].items()))),
          "## Product kernels of the GPU configs\n", hdr]
    for role, pat in PRODUCT:
        for r in rows:
            if pat in r["name"]:
                md.append("| %s | `%s` | %s |\n" % (role, r["name"], " | ".join(str(r.get(c, "")) for c in cols)))
    md.append("\n## Device functions that carry a product kernel's hot loop (read off the ISA)\n\n"
              "| role | function | highest VGPR | scratch instructions (whole function) | hot loop: instructions | fp64 VALU | MFMA | scratch | v_readlane / v_writelane |\n|---|---|---|---|---|---|---|---|---|\n")
    for f in fns:
        md.append("| %s | `%s` | %d | %d | %d | %d | %d | %d | %d |\n" % (f["role"], f["name"], f["highest_vgpr"], f["scratch_instructions"],
                  f["hot_loop_instructions"], f["hot_loop_fp64"], f["hot_loop_mfma"], f["hot_loop_scratch"], f["hot_loop_readlane_writelane"]))
    md.append("\n## Where a product kernel's scratch is (read off the ISA)\n\nThe table above gives a kernel's WORST-CASE stack over everything it can "
              "call; these are the `scratch_*` instructions of the kernel body and of each real (noinline) function it calls -- a config executes "
              "the body and ONE of the callee instantiations.\n\n| role | kernel / callee | instructions | `scratch_*` | of them inside a loop |\n|---|---|---|---|---|\n")
    for c in calls:
        if not c["callees"] and not c["scratch_instructions"]:
            continue
        md.append("| %s | `%s` | %d | %d | %d |\n" % (c["role"], c["kernel"], c["instructions"], c["scratch_instructions"], c["scratch_instructions_inside_a_loop"]))
        for f in c["callees"]:
            md.append("| | calls `%s` | %d | %d | |\n" % (f["name"], f["instructions"], f["scratch_instructions"]))
    md.append("\n## Every kernel with scratch\n\n" + hdr)
    for r in sorted(rows, key=lambda r: -int(r.get("scratch_bytes_per_lane", 0) or 0)):
        if r.get("scratch_bytes_per_lane", 0):
            md.append("| | `%s` | %s |\n" % (r["name"], " | ".join(str(r.get(c, "")) for c in cols)))
    open(os.path.join(ROOT, "profiles", "%s_resources.md" % args.tag), "w").write("".join(md))
    print("".join(md))


if __name__ == "__main__":
    main()
