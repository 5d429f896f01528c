"""Counting comparison of sweep outputs against the oracle's (oracle layout: pair_f [M,A,NPF], pair_i [M,A,NPI],
lists [M,A,NL,T-1], cost [M,NC], safe [M]).

TEST / BENCH INFRASTRUCTURE ONLY: used by tests/ and by bench.py's parity leg; the product never imports it.
Nothing here asserts -- `compare` returns the worst deviations and the number of integer mismatches, the caller
decides (tests assert, the bench line prints them).
"""
import numpy as np

from . import fo_oracle as O

PAIR_FLOATS = ("dce", "ttc", "ttce", "max_ego_risk", "max_obst_risk", "max_obst_harm_with_cp", "max_ego_harm",
               "max_obst_harm", "max_collision_probability")
COST_FLOATS = ("wttc", "min_dce", "max_ego_risk_all", "max_obst_risk_all", "max_ego_harm_all", "max_obst_harm_all",
               "max_collision_probability_all", "max_obst_harm_with_cp_all", "min_ttce")


def _dev(a, b):
    """(worst |a - b| over the finite entries of a, number of entries whose nan / inf pattern differs)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    bad = int((np.isnan(a) != np.isnan(b)).sum() + (np.isinf(a) != np.isinf(b)).sum())
    fin = np.isfinite(a) & np.isfinite(b)
    return (float(np.abs(a[fin] - b[fin]).max()) if fin.any() else 0.0), bad


def compare(ref, got, atol=1e-9):
    """ref: oracle.sweep output (with lists); got: the same keys from the implementation under test.
    Returns dict(float_max_abs_err, list_max_abs_err, list_err_beyond_f32_rounding, int_mismatches, pattern_mismatches, pairs).

    Plateau rule (as in tests/test_sweep_gpu.py): where several samples lie within 2 atol of a pair's largest collision
    probability, float noise decides np.argmax's "first maximum" (hr.py:81) and harm_with_cp = obst_harm[argmax cp]
    follows; on those pairs harm_with_cp is checked against the oracle's harm at the index `got` picked, and argmax
    indices count as mismatches only where the maximum is unique and numerically significant."""
    PF, PI, C, L = O.PF, O.PI, O.COST, O.LST
    worst, lworst, lbeyond, imis, pmis = 0.0, 0.0, 0.0, 0, 0
    lists = ref.get("lists")
    have_lists = lists is not None and lists.shape[-1] > 0
    plateau = np.zeros(ref["pair_f"].shape[:2], dtype=bool)
    if have_lists:
        cpv, mxc = lists[:, :, L["cp"], :], ref["pair_f"][..., PF["max_collision_probability"]]
        plateau = (np.nan_to_num(mxc) > 0.01 + atol) & ((np.abs(cpv - mxc[..., None]) <= 2 * atol).sum(axis=-1) > 1)
        if plateau.any():
            gi = got["pair_i"][..., PI["cp_argmax"]].astype(np.int64)
            oh = np.take_along_axis(lists[:, :, L["obst_harm"], :], gi[..., None], axis=-1)[..., 0]
            hw = got["pair_f"][..., PF["max_obst_harm_with_cp"]]
            worst = max(worst, float(np.abs(hw - oh)[plateau].max()))
    for name in PAIR_FLOATS:
        a, b = ref["pair_f"][..., PF[name]], got["pair_f"][..., PF[name]]
        if name == "max_obst_harm_with_cp" and plateau.any():
            a, b = np.where(plateau, 0.0, a), np.where(plateau, 0.0, b)
        d, bad = _dev(a, b)
        worst, pmis = max(worst, d), pmis + bad
    imis += int((ref["pair_i"][..., PI["time_dce"]] != got["pair_i"][..., PI["time_dce"]]).sum())
    imis += int((ref["pair_i"][..., PI["hr_valid"]] != got["pair_i"][..., PI["hr_valid"]]).sum())
    if have_lists:
        for idx_name, lst, mx in (("max_obst_risk_index", L["obst_risk"], "max_obst_risk"),
                                  ("cp_argmax", L["cp"], "max_collision_probability")):
            ri, gi = ref["pair_i"][..., PI[idx_name]], got["pair_i"][..., PI[idx_name]]
            vals, mxv = lists[:, :, lst, :], ref["pair_f"][..., PF[mx]]
            picked = np.take_along_axis(vals, gi[..., None].astype(np.int64), axis=-1)[..., 0]
            imis += int((~(np.isnan(mxv) | (np.abs(picked - mxv) <= atol))).sum())      # points at a non-maximum
            sig = (np.nan_to_num(mxv) > 1e-9) & ((np.abs(vals - mxv[..., None]) <= 2 * atol).sum(axis=-1) == 1)
            imis += int((ri[sig] != gi[sig]).sum())
        if got.get("lists") is not None:
            d, bad = _dev(lists, got["lists"])
            lworst, pmis = max(lworst, d), pmis + bad
            if got["lists"].dtype == np.float32:
                # float32 STORAGE of float64 results: how far the float64 value behind an entry can have been from the
                # oracle's -- the deviation beyond half a float32 ulp of the oracle's value (what the rounding explains)
                g64 = got["lists"].astype(np.float64)
                fin = np.isfinite(lists) & np.isfinite(g64)
                if fin.any():
                    half = 0.5 * np.spacing(np.abs(lists[fin]).astype(np.float32)).astype(np.float64)
                    lbeyond = max(lbeyond, float(np.maximum(np.abs(g64[fin] - lists[fin]) - half, 0.0).max()))
    for name in COST_FLOATS:
        a, b = ref["cost"][:, C[name]], got["cost"][:, C[name]]
        if name == "max_obst_harm_with_cp_all":
            keep = ~plateau.any(axis=1)
            a, b = a[keep], b[keep]
        d, bad = _dev(a, b)
        worst, pmis = max(worst, d), pmis + bad
    for name in ("argmin_dce", "argmin_ttc", "safe"):
        imis += int((ref["cost"][:, C[name]] != got["cost"][:, C[name]]).sum())
    imis += int((np.asarray(ref["safe"]) != np.asarray(got["safe"])).sum())
    return {"float_max_abs_err": worst, "list_max_abs_err": lworst, "list_err_beyond_f32_rounding": lbeyond,
            "int_mismatches": imis, "pattern_mismatches": pmis, "pairs": int(ref["pair_f"].shape[0] * ref["pair_f"].shape[1])}


def merge(acc, part):
    if acc is None:
        return dict(part)
    return {"float_max_abs_err": max(acc["float_max_abs_err"], part["float_max_abs_err"]),
            "list_max_abs_err": max(acc["list_max_abs_err"], part["list_max_abs_err"]),
            "list_err_beyond_f32_rounding": max(acc.get("list_err_beyond_f32_rounding", 0.0), part.get("list_err_beyond_f32_rounding", 0.0)),
            "int_mismatches": acc["int_mismatches"] + part["int_mismatches"],
            "pattern_mismatches": acc["pattern_mismatches"] + part["pattern_mismatches"],
            "pairs": acc["pairs"] + part["pairs"]}
