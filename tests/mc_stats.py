"""Monte-Carlo error of chain averages (test helper): integrated autocorrelation time by Sokal's automatic windowing, effective sample
size, and the z-scores of two chains' means / variances in units of their combined Monte-Carlo error sigma / sqrt(ESS)."""
import numpy as np


def _autocorr(x):
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    d = x - x.mean()
    nfft = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(d, nfft)
    acf = np.fft.irfft(f * np.conj(f), nfft)[:n]
    if acf[0] <= 0:
        return np.ones(1)
    return acf / acf[0]


def tau_int(x, c=5.0):
    """Integrated autocorrelation time 1 + 2 sum_{t>=1} rho_t, summed up to the first window M with M >= c tau(M) (Sokal 1989)."""
    rho = _autocorr(x)
    taus = 2.0 * np.cumsum(rho) - 1.0
    m = np.arange(taus.size)
    ok = np.flatnonzero(m >= c * taus)
    w = ok[0] if ok.size else taus.size - 1
    return max(float(taus[w]), 1.0)


def ess(x):
    return len(x) / tau_int(x)


def mc_error_of_mean(x):
    return float(np.std(x) / np.sqrt(ess(x)))


def compare_chains(a, b):
    """a, b: [n x Nvars] samples of the same target from two samplers.  Returns (z_mean, z_var, ess_a, ess_b) per variable: differences of
    the means (of the variances) in units of the combined Monte-Carlo error; the variance's error from the chain of squared deviations."""
    a, b = np.asarray(a), np.asarray(b)
    nv = a.shape[1]
    zm, zv, ea, eb = np.zeros(nv), np.zeros(nv), np.zeros(nv), np.zeros(nv)
    for k in range(nv):
        xa, xb = a[:, k], b[:, k]
        ea[k], eb[k] = ess(xa), ess(xb)
        se = np.sqrt(xa.var() / ea[k] + xb.var() / eb[k])
        zm[k] = (xa.mean() - xb.mean()) / se if se > 0 else 0.0
        qa, qb = (xa - xa.mean()) ** 2, (xb - xb.mean()) ** 2
        sev = np.sqrt(qa.var() / ess(qa) + qb.var() / ess(qb))
        zv[k] = (qa.mean() - qb.mean()) / sev if sev > 0 else 0.0
    return zm, zv, ea, eb
