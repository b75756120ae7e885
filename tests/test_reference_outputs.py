"""Row N3 pinned on files the REFERENCE wrote (tests/golden/gaussfit_10280410/, copied as data from
tools/convert_fit2prior_table/test_data/10280410_Gaussfit/): params.hdr (Outputs::write_bin_params, outputs.cpp:1244-1333), the three
restore files (Outputs::write_buffer_restore, outputs.cpp:863-1025) and evidence.txt (Diagnostics::evidence_calc + write_evidence,
diagnostics.cpp:980-1066).  Readers must return the printed numbers; writers must reproduce the files' layout line for line.  No GPU."""
import ctypes as C
import os
import re

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gaussfit_10280410")
ROOT = os.path.join(GOLD, "10280410_Gaussfit_restore_A_")
NAMES = ["H1", "tc1", "H2", "tc2", "p2", "B0", "Amax", "numax", "Gauss_sigma"]


def _numbers_after(path, key):
    """Independent (python) reading of the block that follows `! key=`: numbers up to the next '!' line, '*' markers skipped."""
    out, on = [], False
    for ln in open(path):
        t = ln.strip()
        if t.startswith("!"):
            k, _, rest = t[1:].partition("=")
            on = k.strip() == key
            if on:
                out += [float(v) for v in rest.split()]
            continue
        if t.startswith("#") or not t:
            on = on and not t.startswith("#")
            continue
        if on and not t.startswith("*"):
            out += [float(v) for v in t.split()]
    return np.array(out)


def _read_restore(L, S, root):
    a, b, c = np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int32), np.zeros(1, dtype=np.int64)
    assert L.tamcmc_outputs_read_restore(root.encode(), S._p(a, S._ip), S._p(b, S._ip), S._p(c, S._i64p), None, None, None, None) == 0
    nc, nv = int(a[0]), int(b[0])
    v, s, m, cv = np.zeros((nc, nv)), np.zeros(nc), np.zeros((nc, nv)), np.zeros((nc, nv, nv))
    assert L.tamcmc_outputs_read_restore(root.encode(), S._p(a, S._ip), S._p(b, S._ip), S._p(c, S._i64p), S._p(v), S._p(s), S._p(m), S._p(cv)) == 0
    return nc, nv, int(c[0]), v, s, m, cv


def test_restore_reader_returns_the_numbers_the_reference_printed(pkg):
    from tamcmc_c_amd import sampler as S
    L = S._rebind()
    nc, nv, it, v, s, m, cv = _read_restore(L, S, ROOT)
    assert (nc, nv, it) == (4, 9, 99999)
    # spot values typed from the files
    assert np.array_equal(v[0], [1110.3, 549.555, 343.932, 21.4733, 1.96029, 104.466, 80.5809, 213.583, 11.907])
    assert np.array_equal(v[3], [1110.04, 549.36, 344.284, 21.4582, 1.95441, 104.207, 80.7842, 213.656, 11.8951])
    assert np.array_equal(s, [0.00234744, 0.00110771, 0.0414778, 0.00270788])
    assert np.array_equal(m[1], [1110.03, 549.694, 344.525, 21.4792, 1.95426, 104.174, 80.5997, 213.587, 11.954])
    assert cv[0, 0, 0] == 4.20452 and cv[0, 8, 8] == 0.0140964 and cv[1, 0, 0] == 5.07612 and cv[0, 3, 4] == -0.000978479
    # every number, against an independent reading of the same files
    assert np.array_equal(v.ravel(), _numbers_after(ROOT + "1.dat", "vars"))
    assert np.array_equal(s, _numbers_after(ROOT + "2.dat", "sigmas"))
    assert np.array_equal(m.ravel(), _numbers_after(ROOT + "2.dat", "mus"))
    assert np.array_equal(cv.ravel(), _numbers_after(ROOT + "3.dat", "covarmats"))
    for k in range(nc):
        assert np.allclose(cv[k], cv[k].T, rtol=1e-5, atol=1e-9)      # they are covariance matrices (6 printed digits)


def test_restore_writer_reproduces_the_reference_layout(pkg, tmp_path):
    """Same comment block, same `! key=` lines in the same order, same `*<chain>` markers; number lines carry the same values (written
    with 17 significant digits instead of the reference's 6 -- the one deliberate difference)."""
    from tamcmc_c_amd import sampler as S
    L = S._rebind()
    nc, nv, it, v, s, m, cv = _read_restore(L, S, ROOT)
    names = (C.c_char_p * nv)(*[n.encode() for n in NAMES])
    out = str(tmp_path / "mine_")
    assert L.tamcmc_outputs_write_restore(out.encode(), nc, nv, it, names, S._p(v), S._p(s), S._p(m), S._p(cv)) == 0
    num = re.compile(r"^[\s0-9eE+\-.]+$")
    for k in "123":
        ref = open(ROOT + k + ".dat").read().splitlines()
        got = open(out + k + ".dat").read().splitlines()
        assert len(ref) == len(got), k
        mean_block = False      # the *_mean blocks hold the reference's buffer averages: same shape here, other numbers
        for a, b in zip(ref, got):
            if a.startswith("!"):
                mean_block = "_mean" in a.partition("=")[0]
            if a.startswith("! sigmas"):                              # key and numbers on one line
                ka, _, ra = a.partition("=")
                kb, _, rb = b.partition("=")
                assert ka == kb and len(ra.split()) == len(rb.split())
                assert mean_block or np.array_equal(np.array(ra.split(), float), np.array(rb.split(), float))
            elif num.match(a) and not a.startswith(("#", "!", "*")):
                assert len(a.split()) == len(b.split())
                assert mean_block or np.array_equal(np.array(a.split(), float), np.array(b.split(), float))
            else:
                assert a == b                                         # comments, keys, chain markers: byte for byte
    # the *_mean blocks of the reference are buffer averages; this build repeats the last values there -- reading its own file back
    assert np.array_equal(_read_restore(L, S, out)[3], v)


def test_params_header_equals_the_reference_file(pkg, tmp_path):
    """The reference's header of a finished run: Nsamples 100000, 4 chains, 99999 samples done, 9 variables + 1 constant (p1 = 4),
    relax / plength of its Gaussian-envelope model (10 entries each).  Written here in three buffers: the header of the LAST
    buffer must be the reference's file byte for byte (cumulative Nsamples_done, outputs.cpp:1268)."""
    from tamcmc_c_amd import sampler as S
    L = S._rebind()
    relax = np.array([1, 1, 0, 1, 1, 1, 1, 1, 1, 1], dtype=np.int32)
    plength = np.ones(10, dtype=np.int32)
    allnames = ["H1", "tc1", "p1", "H2", "tc2", "p2", "B0", "Amax", "numax", "Gauss_sigma"]
    inputs = np.array([1110.0, 549.0, 4.0, 344.0, 21.4, 1.96, 104.0, 80.5, 213.5, 11.9])
    names = (C.c_char_p * 10)(*[n.encode() for n in allnames])
    rng = np.random.default_rng(5)
    nc, nv = 4, 9
    root = str(tmp_path / "10280410_Gaussfit_A_")
    chunks = [40000, 40000, 19999]
    allsmp = rng.standard_normal((sum(chunks), nc, nv))
    allst = rng.standard_normal((sum(chunks), nc, 3))
    done = 0
    for k, n in enumerate(chunks):
        smp = np.ascontiguousarray(allsmp[done:done + n])
        stt = np.ascontiguousarray(allst[done:done + n])
        assert L.tamcmc_outputs_write_params(root.encode(), S._p(smp), n, nc, nv, 100000, S._p(relax, S._ip), S._p(plength, S._ip), 10, 10,
                                             S._p(inputs), names, int(k > 0)) == 0
        assert L.tamcmc_outputs_write_stat_criteria(root.encode(), S._p(stt), n, nc, int(k > 0)) == 0
        done += n
        assert f"! Nsamples_done={done}\n" in open(root + "params.hdr").read()          # cumulative after every buffer
        assert f"! Nsamples_done={done}\n" in open(root + "stat_criteria.hdr").read()
    assert open(root + "params.hdr").read() == open(os.path.join(GOLD, "10280410_Gaussfit_A_params.hdr")).read()
    st_hdr = open(root + "stat_criteria.hdr").read().splitlines()
    assert st_hdr[3] == "! Nchains= 4"
    assert st_hdr[4] == "! labels= " + "".join(f"{lab}[{i}]   " for lab in ("logLikelihood", "logPrior", "logPosteriors") for i in range(4))
    # what the reference's tools do (getstats.cpp:156): read exactly Nsamples_done rows of a chain
    back = S.read_params(root, 2)
    assert back.shape == (99999, nv) and np.array_equal(back, allsmp[:, 2, :])
    raw = np.fromfile(root + "stat_criteria.bin", dtype="<f8").reshape(99999, 3, nc)
    assert np.array_equal(raw[:, 1, :], allst[:, :, 1])


def _evidence_rows():
    rows, beta, k = [], None, None
    for ln in open(os.path.join(GOLD, "10280410_Gaussfit_A_evidence.txt")):
        if ln.startswith("! beta="):
            beta = np.array(ln.split("=")[1].split(), float)
        elif ln.startswith("! interpolation_factor="):
            k = int(ln.split("=")[1])
        elif not ln.startswith("#") and ln.strip():
            rows.append(np.array(ln.split(), float))
    return beta, k, np.array(rows)


def test_evidence_known_answers_from_the_reference_run(pkg, oracle, tmp_path):
    """20 lines of the reference's evidence file: <logL> per temperature -> evidence (half-sample parabola resampling to 1000 x Nchains
    points, plain average).  Product routine and oracle must return the printed evidence (10 significant digits in, 10 out) and the
    product writer must reproduce the file byte for byte."""
    from tamcmc_c_amd import sampler as S
    beta, k, rows = _evidence_rows()
    assert np.array_equal(beta, [1, 0.25, 0.0625, 0.015625]) and k == 1000 and rows.shape == (20, 6)
    T = 1.0 / beta
    out = tmp_path / "evidence.txt"
    for i, r in enumerate(rows):
        Lb = r[1:5]
        ev, b2, Lb2, _, _ = S.evidence(T, np.repeat(Lb[None, :, None], 3, axis=2), k)      # one "sample" whose logL row is L_beta
        assert np.array_equal(b2, beta) and np.array_equal(Lb2, Lb)
        assert abs(ev - r[5]) <= 2e-9 * abs(r[5]), (i, ev, r[5])
        ev_o = oracle.evidence(T, Lb[None, :], k)[0]
        assert abs(ev_o - r[5]) <= 2e-9 * abs(r[5]) and abs(ev_o - ev) <= 1e-13 * abs(ev)
        L = S._rebind()
        assert L.tamcmc_outputs_write_evidence(str(out).encode(), int(r[0]), 4, S._p(beta), S._p(Lb), k, r[5], int(i == 0)) == 0
    assert open(out).read() == open(os.path.join(GOLD, "10280410_Gaussfit_A_evidence.txt")).read()


def test_acceptance_file_of_a_finished_run(pkg, tmp_path):
    """`10280410_Gaussfit_A_acceptance.txt`: the acceptance diagnostic the reference appends one line to per buffer
    (Outputs::write_txt_acceptance, outputs.cpp:747-790; x = (Ncopy + 0.5) Nbuffer + Nsamples_init, :1838; rates = moved flags of the
    buffer / Nbuffer, :1834-1837).  The reader returns the printed numbers; the writer, fed with them, reproduces the file byte for
    byte (header, Eigen's padding of the rate columns to the widest entry); the x axis is the one reject_rate builds."""
    from tamcmc_c_amd import sampler as S
    src = os.path.join(GOLD, "10280410_Gaussfit_A_acceptance.txt")
    x, r = S.read_acceptance(src)
    assert r.shape == (20, 4) and x[0] == 2500 and x[-1] == 97500
    assert np.array_equal(r[0], [0.237, 0.2316, 0.2764, 0.2348]) and np.array_equal(r[9], [0.2384, 0.2268, 0.26, 0.2384])
    Nbuffer = 5000
    assert np.array_equal(x, (np.arange(20) + 0.5) * Nbuffer + 0)                      # outputs.cpp:1838 with Nsamples_init = 0
    assert np.all((r * Nbuffer - np.round(r * Nbuffer)) ** 2 < 1e-12)                   # counts of moved flags over the buffer length
    out = str(tmp_path / "acc.txt")
    for k in range(20):
        S.write_acceptance(out, x[k], r[k], first=(k == 0))
    assert open(out, "rb").read() == open(src, "rb").read()
    # coldest chain at the target of the adaptation (config_default.cfg: 0.234) as the finished run left it
    assert abs(r[:, 0].mean() - 0.234) < 0.01
