"""examples/c_abi_linreg.c: the C-ABI used from plain C (gcc, host pointers, no Python / torch in the consumer).
CPU: it compiles against include/fmcmc_amd.h, links the shared library and fails loudly without a GPU.
GPU: its output file equals the oracle bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "fmcmc_amd", "lib")


def _build(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libfmcmc_amd.so")):
        pytest.skip("libfmcmc_amd.so is not built (python -m fmcmc_amd.build)")
    exe = str(tmp_path / "c_abi_linreg")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_linreg.c"), "-o", exe, "-L" + LIBDIR, "-lfmcmc_amd",
                    "-Wl,-rpath," + LIBDIR], check=True)
    return exe


def _write_input(path, X, y, init, scale, nsteps, burnin, thin, seed):
    n, p = X.shape
    C, k = init.shape
    with open(path, "wb") as f:
        f.write(np.array([n, p, C, k, nsteps, burnin, thin, seed], np.int64).tobytes())
        f.write(np.ascontiguousarray(X.T).tobytes())          # [p][n] = R's column-major n x p matrix
        f.write(np.ascontiguousarray(y).tobytes())
        f.write(np.ascontiguousarray(init).tobytes())
        f.write(np.ascontiguousarray(scale).tobytes())


def _run(exe, fin, fout):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    return subprocess.run([exe, fin, fout], capture_output=True, text=True, env=env, timeout=300)


def _case():
    from conftest import synth_linreg
    X, y = synth_linreg(1000, 2, 321)
    rng = np.random.default_rng(5)
    init = np.array([0.0, 0.0, 0.0, float(np.std(y))])[None, :] + 0.1 * rng.standard_normal((6, 4))
    init[:, -1] = np.abs(init[:, -1])
    return X, y, init, np.full(4, 0.05)


def test_c_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    exe = _build(tmp_path)
    X, y, init, scale = _case()
    _write_input(str(tmp_path / "in.bin"), X, y, init, scale, 200, 10, 2, 1215)
    r = _run(exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"))
    assert r.returncode == 3 and "no CPU fallback" in r.stderr
    assert not os.path.exists(str(tmp_path / "out.bin"))


@pytest.mark.gpu
def test_c_example_output_equals_the_oracle(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    exe = _build(tmp_path)
    X, y, init, scale = _case()
    nsteps, burnin, thin, seed = 400, 20, 3, 1215
    _write_input(str(tmp_path / "in.bin"), X, y, init, scale, nsteps, burnin, thin, seed)
    r = _run(exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin"))
    assert r.returncode == 0, r.stderr
    assert "kernel lat1" in r.stdout          # (a handful of chains: the latency form, one chain per workgroup)
    ro = O.run(O.Model(O.FAM_LINREG, X, y), O.Kernel(O.K_NORMAL, 4, scale=scale), init, nsteps=nsteps, burnin=burnin,
               thin=thin, seed=seed)
    C, k, S = 6, 4, (nsteps - burnin) // thin
    raw = open(str(tmp_path / "out.bin"), "rb").read()
    samples = np.frombuffer(raw, np.float64, C * k * S).reshape(C, k, S)
    logpost = np.frombuffer(raw, np.float64, C * S, offset=8 * C * k * S).reshape(C, S)
    acc = np.frombuffer(raw, np.int64, C, offset=8 * (C * k * S + C * S))
    assert np.array_equal(samples.view(np.uint64), np.ascontiguousarray(ro.samples_cks).view(np.uint64))
    assert np.array_equal(logpost.view(np.uint64), np.ascontiguousarray(ro.logpost).view(np.uint64))
    assert np.array_equal(acc, ro.accept_count)
