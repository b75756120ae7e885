"""examples/fit_star.cpp: one star fitted from C++ through the C ABI alone (include/tamcmc_*.h), the reference's files in, the
reference's output formats out.  On a GPU its samples must be those of the Python-driven run with the same settings."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "examples", "fit_star")
ARGS = ["local", os.path.join(GOLD, "TF_3443483_local-v3.model"), os.path.join(GOLD, "TF_3443483_local-v3_slice1.data"),
        os.path.join(GOLD, "sampler_test.cfg"), os.path.join(GOLD, "errors_test.cfg")]


def test_example_host_reads_the_files_and_refuses_to_run_without_a_gpu(pkg, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert os.path.exists(EXE), "built by __graft_entry__.build() (tamcmc-c_amd/Makefile)"
    r = subprocess.run([EXE] + ARGS + [str(tmp_path / "out_"), "0"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "model_MS_local_basic (id 11), 16 parameters (9 free), 973 bins" in r.stdout     # the input front end ran
    assert "hip_create" in r.stderr and "no CPU fallback" in r.stderr and "(-7)" in r.stderr   # TAMCMC_ERR_NO_DEVICE
    r = subprocess.run([EXE, "nonsense"] + ARGS[1:] + [str(tmp_path / "out_")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2


@pytest.mark.gpu
def test_example_host_matches_the_python_driven_run(pkg, tmp_path):
    from tamcmc_c_amd import inputs, sampler
    root = str(tmp_path / "star_")
    r = subprocess.run([EXE] + ARGS + [root, "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "accepted moves of the coldest chain" in r.stdout and "Frequency_l" in r.stdout
    for f in ("params.hdr", "params_chain-0.bin", "params_chain-3.bin", "stat_criteria.bin", "restore_1.dat", "restore_2.dat", "restore_3.dat", "evidence.txt"):
        assert os.path.exists(root + f), f
    # the same fit driven from Python: same files, same settings, same seed -> the same samples, bit for bit
    star, inp = inputs.load_local_star(ARGS[1], ARGS[2], 0)
    cfg = inputs.Cfg(ARGS[3])
    kw, out = cfg.sampler_kwargs()
    cfg.close()
    ctx = pkg.HipContext(0, precision=pkg.PRECISION_FAST)
    ctx.set_spectrum(star.x, star.y)
    s = pkg.Sampler(ctx, star, engine="device", seed=20240229, init_errors=inputs.init_errors(ARGS[4], star), **kw)
    s.run(kw["Nt_learn"][-1], record=False)
    smp, stt = s.run(out["Nsamples"], stats=True)
    for m in range(kw["nchains"]):
        assert np.array_equal(sampler.read_params(root, m), smp[:, m, :]), m
    st = np.fromfile(root + "stat_criteria.bin", dtype="<f8").reshape(out["Nsamples"], 3, kw["nchains"])
    assert np.array_equal(st[:, 0, :], stt[:, :, 0]) and np.array_equal(st[:, 2, :], stt[:, :, 2])
    ev, _, Lb, _, _ = sampler.evidence(kw["lambda_temp"] ** np.arange(kw["nchains"]), stt, 1000)
    line = [ln for ln in open(root + "evidence.txt") if not ln.startswith(("#", "!"))][0].split()
    assert int(line[0]) == out["Nsamples"] and np.isclose(float(line[-1]), ev, rtol=1e-9)
    mean = smp[:, 0, :].mean(0)
    row = [ln for ln in r.stdout.splitlines() if ln.startswith("Frequency_l")][0].split()
    fidx = [i for i, k in enumerate(star.index_to_relax) if star.names[k] == "Frequency_l"][0]
    assert np.isclose(float(row[1]), mean[fidx], rtol=1e-7)
    s.close()
    ctx.close()


@pytest.mark.gpu
def test_example_host_fits_the_reference_red_giant(pkg, tmp_path):
    """The same C++ host on the reference's example red giant (KIC 10722175, red-giant `.model` dialect, io_asymptotic priors): the
    host-driven engine with the mixed-mode solver on the device, output files in the reference's formats."""
    from tamcmc_c_amd import sampler
    root = str(tmp_path / "rgb_")
    args = ["asymptotic", os.path.join(GOLD, "RGB_10722175.model"), os.path.join(GOLD, "RGB_10722175.data"),
            os.path.join(GOLD, "sampler_rgb_test.cfg"), os.path.join(GOLD, "errors_rgb_test.cfg"), root]
    r = subprocess.run([EXE] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert "model_RGB_asympt_aj_AppWidth_HarveyLike_v4 (id 25), 92 parameters (54 free), 6099 bins" in r.stdout
    cold = sampler.read_params(root, 0)
    assert cold.shape == (120, 54) and np.isfinite(cold).all() and (cold[1:] != cold[:-1]).any()
    st = np.fromfile(root + "stat_criteria.bin", dtype="<f8").reshape(120, 3, 4)
    assert np.isfinite(st).all() and st[:, 2, 0].mean() > st[0, 2, 0] - 50.0
    hdr = open(root + "params.hdr").read()
    assert "! Nchains= 4" in hdr and "DP1" in hdr and "ferr_bias" in hdr
    for f in ("restore_1.dat", "restore_3.dat", "evidence.txt"):
        assert os.path.exists(root + f)
