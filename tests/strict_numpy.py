"""numpy emulation of the STRICT kernel's per-bin arithmetic on a multiplet table (test helper).
Evaluates the table the product's host builder emitted with the reference's operation order, so the
host logic (parameter unpack, windows, nu_nlm, H*V) can be checked bit-for-bit against the oracle
without a GPU."""
import numpy as np


def eval_table(mults, noise_abs, nharvey, x):
    model = np.zeros(x.size)
    for r in mults:
        l, i0, i1 = int(r["l"]), int(r["i0"]), int(r["i1"])
        xl = x[i0:i1]
        g2 = r["gamma"] * r["gamma"]
        res = np.zeros(xl.size)
        for k in range(2 * l + 1):
            d = xl - r["nu"][k]
            prof = d * d
            prof = 4.0 * prof / g2
            inv = 1.0 / (1.0 + prof)
            if r["asym"] == 0.0:
                res = res + r["hv"][k] * inv
            else:
                c2 = 0.5 * r["gamma"] * r["asym"] / r["fc"]
                t = 1.0 + r["asym"] * (xl / r["fc"] - 1.0)
                asy = t * t + c2 * c2
                res = res + r["hv"][k] * (asy * inv)
        model[i0:i1] = model[i0:i1] + res
    for h in range(nharvey):
        if noise_abs[3 * h + 1] != 0:
            t = ((1e-3 * noise_abs[3 * h + 1]) * x) ** noise_abs[3 * h + 2]
            model = model + noise_abs[3 * h] * (1.0 / (t + 1.0))
    return model + noise_abs[-1]
