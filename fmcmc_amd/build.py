"""Builds the HIP engine in-tree: fmcmc_amd/lib/libfmcmc_amd.so (gfx950 only).

The library is several translation units (csrc/mh_engine.hip = C-ABI + kernel selection, csrc/k_*.hip = one kernel family
each, csrc/gelman.hip) compiled in parallel into build/*.o and linked by hipcc; only the units whose sources changed are
recompiled."""
import concurrent.futures
import glob
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libfmcmc_amd.so")
OBJDIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-value", "-Wno-unused-result"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def deps():
    return sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))


def unit_deps(src, _seen=None):
    """the files a translation unit includes (quoted includes, followed recursively)"""
    import re
    seen = _seen if _seen is not None else set()
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), flags=re.M):
        f = os.path.normpath(os.path.join(os.path.dirname(src), inc))
        if os.path.exists(f) and f not in seen:
            seen.add(f)
            unit_deps(f, seen)
    return seen


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in sources() + deps())


def _compile(src, obj, flags, verbose):
    cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.time()
    subprocess.check_call(cmd)
    return os.path.basename(src), time.time() - t0


def build(force=False, verbose=False, extra_flags=(), out=None, jobs=None):
    """extra_flags / out: diagnostic variants next to the product library, loaded through FMCMC_AMD_LIB (e.g. -DFMCMC_STAMP ->
    lib/libfmcmc_amd_stamp.so, tools/stamp_wide.py); their objects go to build/<name of out>/.  Returns the library path;
    build.last_times holds the compile seconds per translation unit of the last call."""
    out = out or OUT
    if not force and not extra_flags and not needs_build():
        return out
    variant = "" if out == OUT else os.path.splitext(os.path.basename(out))[0]
    objdir = os.path.join(OBJDIR, variant) if variant else OBJDIR
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flags = FLAGS + list(extra_flags)
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        dep_t = max(os.path.getmtime(f) for f in [src] + sorted(unit_deps(src)))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < dep_t:
            todo.append((src, obj))
    # (longest first: the units that instantiate the most kernels decide the wall time when they start last)
    HEAVY = ("k_lat_l2a", "k_lat_l1a", "k_wide.", "k_spec_a", "k_lat2a", "k_lat1a", "k_lat2c", "k_lat_l2b", "k_lat_l1b", "k_lat1c", "k_mfma_ad", "k_spec_l2", "k_spec_w1", "k_spec_w3", "k_spec_lw1", "k_spec_lw2", "k_lat_l3a", "k_lat_l3b", "k_lat3a", "k_lat3b", "k_logit1", "k_lat_l2c", "k_lat_l1c", "k_lat2b", "k_lat1b", "k_lat2d", "k_lat1d", "k_logit2", "k_mfma2", "k_mfma1", "k_general", "k_logit0", "k_spec_r")
    def cost(so):
        b = os.path.basename(so[0])
        for i, h in enumerate(HEAVY):
            if b.startswith(h):
                return i
        return len(HEAVY)
    todo.sort(key=cost)
    jobs = jobs or int(os.environ.get("FMCMC_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    t0 = time.time()
    times = {}
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        for name, dt in ex.map(lambda so: _compile(so[0], so[1], flags, verbose), todo):
            times[name] = dt
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    times["_wall"] = time.time() - t0
    build.last_times = times
    if verbose:
        print("compile seconds per unit:", {k: round(v, 1) for k, v in sorted(times.items())}, flush=True)
    return out


build.last_times = {}

if __name__ == "__main__":
    build(force=True, verbose=True)
