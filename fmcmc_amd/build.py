"""Builds the HIP engine in-tree: fmcmc_amd/lib/libfmcmc_amd.so (gfx950 only)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "mh_engine.hip"), os.path.join(HERE, "csrc", "gelman.hip")]
DEPS = [os.path.join(ROOT, "include", f) for f in ("fmcmc_amd.h", "fmh_detmath.h", "fmh_philox.h")] + \
       [os.path.join(HERE, "csrc", f) for f in ("mh_common.hpp", "mh_streamed.hpp", "mh_pipe.hpp", "mh_mfma.hpp", "mh_mfma_rep.hpp", "mh_spec.hpp", "mh_wide2.hpp", "mh_mfma_ad.hpp", "mh_bigk.hpp")] + \
       [os.path.join(ROOT, "include", "fmh_logit_tab.h")]
OUT = os.path.join(HERE, "lib", "libfmcmc_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wno-unused-value", "-Wno-unused-result"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in SRC + DEPS)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """extra_flags / out: diagnostic variants next to the product library, loaded through FMCMC_AMD_LIB: -DFMCMC_STAMP ->
    lib/libfmcmc_amd_stamp.so (tools/stamp_wide.py); -DFMCMC_AB -> lib/libfmcmc_amd_ab.so, which also carries the A/B partners
    of the product kernels (mh_sweep_pipe, mh_sweep_mfmar, the stamped MFMA instantiations: knobs spec=0 / owners=0 / mode=8)."""
    out = out or OUT
    if not force and not extra_flags and not needs_build():
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = [s for s in SRC if os.path.exists(s)]
    cmd = [HIPCC] + FLAGS + list(extra_flags) + src + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force=True, verbose=True)
