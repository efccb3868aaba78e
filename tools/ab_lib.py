"""A/B of two builds of the library on one box: us per MH iteration of a perf-guard case under each (FMCMC_AMD_LIB), alternating.
   python tools/ab_lib.py <libA.so> <libB.so> [case=c3@256 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
cases = sys.argv[3:] or ["c3@256"]
code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import test_gpu_perf_guard as g\n"
        "for c in sys.argv[1:]:\n    us, k, _ = g._measure(c); print(c, k, round(us, 3), flush=True)\n") % (ROOT, ROOT)
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for lib in libs:
        env = dict(os.environ, FMCMC_AMD_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "-c", code] + cases, env=env, capture_output=True, text=True)
        print(os.path.basename(lib), "|", " ; ".join(out.stdout.strip().split("\n")), out.stderr.strip()[-200:] if out.returncode else "", flush=True)
