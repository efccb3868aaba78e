"""shim/fmcmc_amd_shim.c -- the `.Call` side of the drop-in -- compiled and EXECUTED without R.

R is not in this image, so the shim is built against tests/rapi_stub/ (a stand-in for the handful of R API calls it makes,
written from "Writing R Extensions") and driven by tests/shim_harness.c the way R would: R_init_fmcmc registers the .Call
table, entry points are looked up by name and arity, arguments are the named lists shim/amd_hook.R builds, every call runs
inside the stub's error context (which also checks the protect stack).

CPU: -Wall -Werror -fsanitize=address,undefined build; argument errors come back as R errors with the reference's own texts
     (/root/reference/inst/tinytest/test-mcmc.R:3-23, R/mcmc.R:501-520, R/kernel.R:9,129-132, R/kernel_normal.R:134-135).
GPU: C_fmcmc_amd_run for kernel_normal and kernel_ram, two consecutive calls (the second continues from the state lists the
     first returned), every returned array equal to the oracle bit for bit; a NaN log-posterior ends in the R error of
     R/mcmc.R:759-765."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "fmcmc_amd", "lib")
STUB = os.path.join(ROOT, "tests", "rapi_stub")


def _build(tmp_path, sanitize):
    if not os.path.exists(os.path.join(LIBDIR, "libfmcmc_amd.so")):
        pytest.skip("libfmcmc_amd.so is not built (python -m fmcmc_amd.build)")
    exe = str(tmp_path / ("shim_harness_san" if sanitize else "shim_harness"))
    flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if sanitize else ["-O2"]
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Werror"] + flags +
                   ["-I" + STUB, "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "shim", "fmcmc_amd_shim.c"), os.path.join(STUB, "rapi_stub.c"),
                    os.path.join(ROOT, "tests", "shim_harness.c"), "-o", exe, "-L" + LIBDIR, "-lfmcmc_amd", "-lm",
                    "-Wl,-rpath," + LIBDIR], check=True)
    return exe


def _run(exe, *args):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    # (leak detection off: the HIP runtime the library links keeps process-lifetime allocations; our own records are freed)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1"
    return subprocess.run([exe] + list(args), capture_output=True, text=True, env=env, timeout=600)


def test_shim_compiles_clean_and_raises_the_references_argument_errors(tmp_path):
    exe = _build(tmp_path, sanitize=True)
    r = _run(exe, "errors")
    assert r.returncode == 0, (r.stdout, r.stderr)
    got = dict(line.split("|", 1) for line in r.stdout.strip().splitlines())
    sys.path.insert(0, ROOT)
    from fmcmc_amd import _abi as abi
    assert got["info"] == "<no error>" and int(got["info_abi"]) == abi.ABI_VERSION
    assert got["ok"] == "<no error>"
    # inst/tinytest/test-mcmc.R:4-17 expects "burnin" / "thin" in the messages of R/mcmc.R:511-520
    assert got["burnin"] == "-burnin- (100) cannot be >= than -nsteps- (100)."
    assert got["run_burnin"] == got["burnin"]
    assert got["thin_negative"] == "-thin- should be >= 1."
    assert got["thin_nsteps"] == "-thin- (100) cannot be > than -nsteps- (100)."
    assert got["nchains"] == "`nchains` must be an integer greater than 1."
    assert got["scale_length"] == "Incorrect length of -scale-."                       # R/kernel.R:9
    assert got["ub_lb"] == "-ub- cannot be <= than -lb-."                               # R/kernel_normal.R:134-135
    assert got["all_fixed"].startswith("The number of parameters to update, i.e. not fixed, cannot be zero.")   # R/kernel.R:129-132
    assert "-seed- must be a whole number" in got["seed_negative"] and "-seed- must be a whole number" in got["seed_na"]
    assert got["fed_z_short"] == "Incorrect length of -fed_z-."


def _write_input(path, X, y, init, scale, nsteps, burnin, thin, seed, kind, guard, nsteps2):
    n, p = X.shape
    C, k = init.shape
    with open(path, "wb") as f:
        f.write(np.array([n, p, C, k, nsteps, burnin, thin, seed, kind, guard, nsteps2], np.int64).tobytes())
        f.write(np.ascontiguousarray(X.T).tobytes())          # [p][n] = R's column-major n x p matrix
        f.write(np.ascontiguousarray(y).tobytes())
        f.write(np.ascontiguousarray(init).tobytes())          # [C][k] = t(initial) as a k x C R matrix
        f.write(np.ascontiguousarray(scale).tobytes())


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["normal", "ram"])
def test_shim_run_equals_the_oracle_bit_for_bit(tmp_path, kind):
    sys.path.insert(0, ROOT)
    from conftest import synth_linreg
    from oracle import oracle as O
    from fmcmc_amd import _abi as abi
    exe = _build(tmp_path, sanitize=False)
    X, y = synth_linreg(1000, 2, 321)
    rng = np.random.default_rng(5)
    C, k = 6, 4
    init = np.array([0.0, 0.0, 0.0, float(np.std(y))])[None, :] + 0.1 * rng.standard_normal((C, k))
    init[:, -1] = np.abs(init[:, -1])
    scale = np.full(k, 0.05 if kind == "normal" else 1.0)
    nsteps, burnin, thin, seed, nsteps2 = 400, 20, 3, 1215, 150
    akind = abi.KERNEL_NORMAL if kind == "normal" else abi.KERNEL_RAM
    _write_input(str(tmp_path / "in.bin"), X, y, init, scale, nsteps, burnin, thin, seed, akind, 1, nsteps2)
    r = _run(exe, "run", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"))
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "call 1: kernel" in r.stdout and "call 2: kernel" in r.stdout
    ok = O.Kernel(O.K_NORMAL, k, scale=scale) if kind == "normal" else O.Kernel(O.K_RAM, k)
    st = O.ChainState(init, ok.kf)
    model = O.Model(O.FAM_LINREG, X, y)
    raw = np.frombuffer(open(str(tmp_path / "out.bin"), "rb").read(), np.float64)
    off = 0
    for (ns, bi) in ((nsteps, burnin), (nsteps2, 0)):
        ro = O.run(model, ok, None, nsteps=ns, burnin=bi, thin=thin, seed=seed, state=st, want_draws=True)
        S = (ns - bi) // thin
        blocks = []
        for cnt in (C * k * S, C * S, C * k * S, C, C * k):
            blocks.append(raw[off:off + cnt]); off += cnt
        samples, logpost, draws, acc, theta0 = blocks
        assert np.array_equal(_bits(samples.reshape(C, k, S)), _bits(ro.samples_cks)), "samples (call with %d steps)" % ns
        assert np.array_equal(_bits(logpost.reshape(C, S)), _bits(ro.logpost))
        assert np.array_equal(_bits(draws.reshape(C, k, S)), _bits(ro.draws_cks))
        assert np.array_equal(acc.astype(np.int64), ro.accept_count)
        assert np.array_equal(_bits(theta0.reshape(C, k)), _bits(st.theta0))
    assert off == raw.size


@pytest.mark.gpu
def test_shim_reports_an_undefined_logposterior_as_the_reference_does(tmp_path):
    """inst/tinytest/test-mcmc.R:19-23: f <- function(i) NaN -> error matching "undefined" (R/mcmc.R:759-765).  Here sigma steps
    below zero without the guard; the message names the step and theta1 ONCE."""
    sys.path.insert(0, ROOT)
    from conftest import synth_linreg
    from fmcmc_amd import _abi as abi
    exe = _build(tmp_path, sanitize=False)
    X, y = synth_linreg(300, 1, 2)
    init = np.tile([0.0, 0.0, 0.05], (2, 1))
    _write_input(str(tmp_path / "in.bin"), X, y, init, np.ones(3), 200, 0, 1, 7, abi.KERNEL_NORMAL, 0, 0)
    r = _run(exe, "run", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"))
    assert r.returncode == 4, (r.stdout, r.stderr)
    msg = [l for l in r.stdout.splitlines() if l.startswith("error|")][0]
    assert "undefined" in msg and "Check either -fun- or the -lb- and -ub- parameters." in msg
    assert msg.count("This error ocurred during step i =") == 1 and "theta1 = c(" in msg
