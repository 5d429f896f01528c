#!/usr/bin/env python3
"""Node counts of the correlation integral (fo_sweep.hip fo_corr_corners): largest absolute error of the n-node
Gauss-Legendre rule over the correlation angle, against a 400-node rule of the same formula (itself equal to scipy's
mvnun to 2e-15, checked at the end), on boxes whose corners lie near the mean and near the diagonals h = +-k where the
integrand is sharpest.  CPU only: `python tools/corr_nodes.py`."""
import numpy as np
from numpy.polynomial.legendre import leggauss
from scipy.special import erf


def box(n, A, B, C, D, rho):
    x, w = leggauss(n)
    asr = np.arcsin(rho)
    s = np.sin(asr * (x + 1) / 2)
    c2 = 1 / (1 - s * s)
    E = lambda h, k: np.exp(-(h * h + k * k - 2 * h * k * s) * c2)
    return 0.25 * (erf(B) - erf(A)) * (erf(D) - erf(C)) + (w * (E(A, C) - E(B, C) - E(A, D) + E(B, D))).sum() * asr / (4 * np.pi)


def main():
    rng = np.random.default_rng(4)
    ns = (6, 8, 12, 16, 20, 24, 32)
    print("rho    " + "".join(f"{n:>10d}" for n in ns))
    for rho in (0.3, 0.5, 0.7, 0.9, 0.95, 0.97, 0.99):
        err = {n: 0.0 for n in ns}
        for _ in range(4000):
            h = rng.uniform(-3, 3)
            k = rng.choice([h, -h, rng.uniform(-3, 3)]) + rng.normal() * rng.choice([0.01, 0.1, 0.5])
            A, B, C, D = h, h + rng.uniform(0.2, 5), k, k + rng.uniform(0.2, 5)
            r = rho * rng.choice([-1, 1])
            ref = box(400, A, B, C, D, r)
            for n in ns:
                err[n] = max(err[n], abs(box(n, A, B, C, D, r) - ref))
        print(f"{rho:<7}" + "".join(f"{err[n]:>10.1e}" for n in ns))
    try:
        from scipy.stats import _mvn
        worst = 0.0
        for _ in range(200):
            A, C = rng.uniform(-2, 0, 2)
            B, D = A + rng.uniform(.5, 2), C + rng.uniform(.5, 2)
            r = rng.uniform(-0.99, 0.99)
            v, _ = _mvn.mvnun(np.array([A, C]) * 2 ** .5, np.array([B, D]) * 2 ** .5, np.zeros(2), np.array([[1, r], [r, 1]]))
            worst = max(worst, abs(v - box(400, A, B, C, D, r)))
        print("400 nodes against scipy mvnun:", f"{worst:.1e}")
    except Exception as ex:   # scipy without the private Fortran wrapper
        print("mvnun not available:", ex)


if __name__ == "__main__":
    main()
