"""Output formats of the kept surface (row N3): params .bin/.hdr, stat_criteria, bin2txt-style summary statistics. No GPU."""
import numpy as np


def test_params_bin_hdr_roundtrip_and_summary(pkg, synth, tmp_path):
    from tamcmc_c_amd import sampler
    star = synth.make_c2_star(nx=512)
    rng = np.random.default_rng(2)
    n, nc, nv = 257, 3, star.nvars
    smp = rng.standard_normal((n, nc, nv)) * np.arange(1, nv + 1) + 10.0
    stt = rng.standard_normal((n, nc, 3))
    root = str(tmp_path / "star_A_")
    sampler.write_outputs(root, star, smp[:100], stt[:100], nsamples_total=n)
    sampler.write_outputs(root, star, smp[100:], stt[100:], append=True)
    for m in range(nc):
        back = sampler.read_params(root, m)
        assert back.shape == (n, nv) and np.array_equal(back, smp[:, m, :])
        raw = np.fromfile(root + f"params_chain-{m}.bin", dtype="<f8")      # raw little-endian doubles [sample][var]
        assert np.array_equal(raw.reshape(n, nv), smp[:, m, :])
    hdr = open(root + "params.hdr").read()
    assert f"! Nchains= {nc}" in hdr and f"! Nvars= {nv}" in hdr and f"! Ncons= {star.params.size - nv}" in hdr
    assert "! plength= " + " ".join(str(v) for v in star.plength) in hdr
    assert "! variable_names=" in hdr and "Frequency_l" in hdr and "! constant_values=" in hdr
    st = np.fromfile(root + "stat_criteria.bin", dtype="<f8").reshape(n, 3, nc)   # logL[chains], logPrior[chains], logPost[chains]
    assert np.array_equal(st[:, 0, :], stt[:, :, 0]) and np.array_equal(st[:, 2, :], stt[:, :, 2])
    mean, med, sd = sampler.params_summary(smp[:, 0, :])
    assert np.allclose(mean, smp[:, 0, :].mean(0), rtol=1e-14) and np.allclose(med, np.median(smp[:, 0, :], axis=0), rtol=1e-15)
    assert np.allclose(sd, smp[:, 0, :].std(0), rtol=1e-13)        # population standard deviation, like the reference tool
    mean2, med2, _ = sampler.params_summary(smp[:256, 0, :])       # even count: median = mean of the two central values
    assert np.allclose(med2, np.median(smp[:256, 0, :], axis=0), rtol=1e-15)
