"""The division-free crossing test of the rule kernels (csrc/fo_spawn_rules.hpp, rl_left_of_crossing) against the quotient form it
replaces, in NumPy float64 with the device's operation order (the translation unit is built with -ffp-contract=off: no fused
multiply-adds).  The kernel decides ``x < xi + (y - yi) (xj - xi) / (yj - yi)`` by the sign of s = (x - xi) d - (y - yi)(xj - xi)
against the sign of d = yj - yi whenever |s| > 2^-48 (|t2| + |m| + |d xi|), and by the quotient itself otherwise.  Claim (DESIGN
section 5, the comment in the header): whenever the sign form decides, it decides like the quotient form -- checked here where
it matters, on points a few units in the last place either side of the computed crossing.  CPU only."""
import numpy as np


def _quotient_form(x, y, xi, yi, xj, yj):
    with np.errstate(all="ignore"):
        return x < xi + ((y - yi) * (xj - xi)) / (yj - yi)


def _sign_form(x, y, xi, yi, xj, yj):
    """(decided, answer) per element -- the arithmetic of rl_left_of_crossing"""
    d = yj - yi
    m = (y - yi) * (xj - xi)
    t2 = (x - xi) * d
    s = t2 - m
    decided = np.abs(s) > 2.0 ** -48 * (np.abs(t2) + np.abs(m) + np.abs(d * xi))
    return decided, (s < 0.0) != (d < 0.0)


def _edges(rng, n, scale):
    xi, xj = rng.uniform(-scale, scale, n), rng.uniform(-scale, scale, n)
    yi = rng.uniform(-scale, scale, n)
    yj = yi + rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-6, np.log10(scale), n)      # short and long, steep and flat edges
    lo, hi = np.minimum(yi, yj), np.maximum(yi, yj)
    y = lo + rng.uniform(0.0, 1.0, n) * (hi - lo)
    ok = (yi > y) != (yj > y)                                                               # the straddle condition of the caller
    return tuple(a[ok] for a in (xi, yi, xj, yj, y))


def test_sign_form_agrees_with_the_quotient_form_wherever_it_decides():
    rng = np.random.default_rng(20240607)
    undecided_near, n_near, n_far, undecided_far = 0, 0, 0, 0
    for scale in (1.0, 50.0, 1e3, 1e5):
        xi, yi, xj, yj, y = _edges(rng, 200_000, scale)
        with np.errstate(all="ignore"):
            xc = xi + ((y - yi) * (xj - xi)) / (yj - yi)
        # (a) points within a few ulps of the computed crossing, both sides and on it
        for k in range(-6, 7):
            x = xc.copy()
            for _ in range(abs(k)):
                x = np.nextafter(x, np.inf if k > 0 else -np.inf)
            want = _quotient_form(x, y, xi, yi, xj, yj)
            dec, got = _sign_form(x, y, xi, yi, xj, yj)
            assert np.array_equal(got[dec], want[dec]), (scale, k)
            undecided_near += int((~dec).sum())
            n_near += len(x)
        # (b) points at a relative distance of 1e-15 ... 1e-9 from the crossing (where the sign form starts to decide)
        for rel in (1e-15, 3e-15, 1e-14, 1e-13, 1e-12, 1e-9):
            x = xc + rng.choice([-1.0, 1.0], len(xc)) * rel * (np.abs(xc) + np.abs(xi) + np.abs(xj - xi))
            want = _quotient_form(x, y, xi, yi, xj, yj)
            dec, got = _sign_form(x, y, xi, yi, xj, yj)
            assert np.array_equal(got[dec], want[dec]), (scale, rel)
        # (c) points anywhere: the sign form decides practically always (the quotient is the rare way)
        x = rng.uniform(-scale, scale, len(xc))
        want = _quotient_form(x, y, xi, yi, xj, yj)
        dec, got = _sign_form(x, y, xi, yi, xj, yj)
        assert np.array_equal(got[dec], want[dec]), scale
        undecided_far += int((~dec).sum())
        n_far += len(x)
    assert undecided_near > 0.5 * n_near          # next to the crossing the quotient decides (the bound is not vacuous)
    assert undecided_far < 1e-6 * n_far + 5       # elsewhere it is not needed


def test_degenerate_inputs_take_the_quotient():
    """NaN and infinite coordinates fail the comparison of the sign form (the kernel then evaluates the quotient as before); a
    point ON a vertex row (y = yi: the product is an exact zero) is decided like the quotient form"""
    nan, inf = np.nan, np.inf
    for args in ((nan, 0.5, 0.0, 0.0, 1.0, 1.0), (0.2, 0.5, nan, 0.0, 1.0, 1.0), (inf, 0.5, 0.0, 0.0, 1.0, 1.0), (0.2, 0.5, 0.0, 0.0, inf, 1.0)):
        dec, _ = _sign_form(*(np.array([a]) for a in args))
        assert not dec[0], args
    x = np.array([-1.0, 0.0, np.nextafter(0.0, 1.0), 1.0])
    y, xi, yi, xj, yj = (np.full(4, v) for v in (0.0, 0.0, 0.0, 3.0, 2.0))      # y == yi, straddled since yj > y
    dec, got = _sign_form(x, y, xi, yi, xj, yj)
    assert np.array_equal(got[dec], _quotient_form(x, y, xi, yi, xj, yj)[dec]) and dec[0] and dec[3]


def test_square_root_free_distance_tests_are_exact():
    """member_idx of the dynamic rule compares squared distances: ``sqrt(d2) <= 12`` as ``d2 <= 144`` and ``sqrt(e2) > 1`` as
    ``e2 > 1 + 2^-52`` (correctly rounded square roots; the reference's comparisons are on the roots, spawn_locator.py:256-262)
    -- exact for every double, checked on the two thousand neighbours of either threshold"""
    d2 = np.full(2001, 144.0)
    e2 = np.full(2001, 1.0)
    for k in range(1, 1001):
        d2[1000 + k], d2[1000 - k] = np.nextafter(d2[1000 + k - 1], np.inf), np.nextafter(d2[1000 - k + 1], -np.inf)
        e2[1000 + k], e2[1000 - k] = np.nextafter(e2[1000 + k - 1], np.inf), np.nextafter(e2[1000 - k + 1], -np.inf)
    assert np.array_equal(np.sqrt(d2) <= 12.0, d2 <= 144.0)
    assert np.array_equal(np.sqrt(e2) > 1.0, e2 > 1.0000000000000002)
    assert (np.sqrt(d2) <= 12.0).sum() == 1001 and (np.sqrt(e2) > 1.0).sum() == 999      # (1 + 2^-52 itself roots to 1)


def test_quotient_free_unit_interval_test_of_the_shadow_wedge():
    """the dynamic rule's shadow test (sight line ego -> point against the obstacle's four sides, spawn_locator.py:264) asks
    ``0 <= tn / den <= 1`` as sign agreement and |tn| <= |den|: a correctly rounded quotient is <= 1 exactly when |tn| <= |den|
    and >= 0 exactly when the signs agree or tn = 0.  Checked on random pairs and on the neighbours of |tn| = |den| and of
    tn = 0 (outside the underflow range: a quotient below 2^-1075 in magnitude rounds to a signed zero, which compares >= 0 --
    the cross products of metre-scale coordinates are exact zeros or many orders above it)"""
    rng = np.random.default_rng(5)
    den = rng.normal(0.0, 1.0, 200_000) * 10.0 ** rng.uniform(-8, 8, 200_000)
    den = den[np.abs(den) > 1e-14]                       # (the kernel's own guard, as in the reference)
    n = len(den)
    tn = np.concatenate((den * rng.uniform(-0.5, 1.5, n), den, -den, np.zeros(n), np.nextafter(den, np.inf), np.nextafter(den, -np.inf),
                         np.copysign(5e-324 * 2.0 ** 60, den), -np.copysign(5e-324 * 2.0 ** 60, den) * (np.abs(den) < 1e3)))
    dd = np.tile(den, 8)
    with np.errstate(all="ignore"):
        q = tn / dd
    want = (q >= 0.0) & (q <= 1.0)
    got = np.where(dd > 0.0, (tn >= 0.0) & (tn <= dd), (tn <= 0.0) & (tn >= dd))
    assert np.array_equal(got, want)
