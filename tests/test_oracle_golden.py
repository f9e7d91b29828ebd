"""Pins the CPU oracle (oracle/wol_oracle.c) to the golden vectors that the reference itself
produced (tests/golden/make_golden.py).  Bit-exact on indices, distances, masks and integer
histograms; q/angles/LSI are compared at 1e-12 (libm vs numpy's SIMD cos may differ by an ulp)
although in this container they agree to the last bit."""
import numpy as np
import pytest

from conftest import EXTRA_CASES, Q_CASES, case_inputs, golden


@pytest.mark.parametrize("name", Q_CASES)
def test_order_param_q_literal(oracle, name):
    g = golden(name)
    sub, pos, box, lo, hi = case_inputs(g)
    q, idx, dist, ang = oracle.getOrderParamq(sub, pos, box, lo, hi, full=True)
    assert np.array_equal(idx, g["nn_idx"])
    assert np.array_equal(dist, g["nn_dist"], equal_nan=True)
    np.testing.assert_allclose(ang, g["ang"], rtol=0, atol=1e-10, equal_nan=True)
    np.testing.assert_allclose(q, g["q"], rtol=0, atol=1e-12)
    assert np.array_equal(oracle.np_histogram(q), g["qhist"])
    assert abs(oracle.np_mean(g["q"]) - float(g["q_mean"])) == 0.0
    assert abs(oracle.np_var(g["q"]) - float(g["q_var"])) == 0.0


@pytest.mark.parametrize("name", [c for c in Q_CASES if c != "nonperiodic_z"])
def test_order_param_q_cells_equals_literal(oracle, name):
    g = golden(name)
    sub, pos, box, lo, hi = case_inputs(g)
    a = oracle.getOrderParamq(sub, pos, box, lo, hi, full=True)
    b = oracle.getOrderParamq(sub, pos, box, lo, hi, full=True, cells=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("name", Q_CASES)
def test_masks(oracle, name):
    g = golden(name)
    sub, pos, box, lo, hi = case_inputs(g)
    m = oracle.nearneighbors(sub, pos, box, lo, hi)
    assert m.flags.f_contiguous and m.dtype == np.int32
    assert int(m.sum()) == int(g["mask_sum"])
    if "sub" not in g.files:
        assert np.array_equal(m, oracle.allnearneighbors(pos, box, lo, hi))
    # the mask row count bounds the number of selected neighbours
    assert np.array_equal(np.minimum(m.sum(axis=1), 4), (g["nn_idx"] >= 0).sum(axis=1))


@pytest.mark.parametrize("name", EXTRA_CASES)
def test_cos_angs_and_lsi(oracle, name):
    g = golden(name)
    sub, pos, box, _, _ = case_inputs(g)
    ca, cn = oracle.getCosAngs(sub, pos, box)
    assert ca.shape == g["cos_ang"].shape and np.array_equal(cn, g["cos_num"])
    np.testing.assert_allclose(ca, g["cos_ang"], rtol=0, atol=1e-10)
    assert np.array_equal(oracle.np_histogram(ca, 500, (0.0, 180.0)), g["ang_hist"])
    li, ln = oracle.getLSI(sub, pos, box)
    assert li.shape == g["lsi"].shape and np.array_equal(ln, g["lsi_num"])
    np.testing.assert_allclose(li, g["lsi"], rtol=0, atol=1e-12)


def test_traj_reduction(oracle):
    g = golden("traj700x3")
    for cells in (False, True):
        mean, var, qh, dh = oracle.tet_order_traj(g["frames"], g["boxes"], cells=cells)
        assert np.array_equal(mean, g["mean"]) and np.array_equal(var, g["var"])
        assert np.array_equal(qh, g["qhist"])
        assert dh.sum() == 4 * 700 * 3


def test_waterlib_known_answers(oracle):
    g = golden("waterlib_kat")
    ang = oracle.tetracosang(g["tet_pos"][0], g["tet_pos"][1:], g["tet_box"])
    np.testing.assert_allclose(ang[np.triu_indices(4, k=1)], g["tet_ang"], rtol=0, atol=1e-10)
    assert np.all(np.abs(g["tet_ang"] - 109.4712206) < 0.1)
    assert np.array_equal(oracle.reimage(g["reimage_in"], np.zeros(3), np.array([10.0] * 3)), g["reimage_out"])
    # anint is round-half-away: +5 -> -5 and -5 -> +5 at exactly L/2
    assert g["reimage_out"][0, 0] == -5.0 and g["reimage_out"][0, 1] == 5.0
    assert np.array_equal(oracle.lsidists(g["tet_pos"][0], g["tet_pos"][1:] + 40.0, g["tet_box"]), g["lsidists"])
    q = oracle.getOrderParamq(g["tet_pos"][:1], g["tet_pos"], g["tet_box"])
    np.testing.assert_allclose(q, g["tet_q"], atol=1e-12)


def test_analytic_cases(oracle):
    # perfect tetrahedron => q == 1 and six angles of acos(-1/3)
    tet = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], float)
    pos = np.vstack([np.zeros(3), tet * 1.6]) + 15.0
    q, idx, dist, ang = oracle.getOrderParamq(pos[:1], pos, np.array([30.0] * 3), full=True)
    assert abs(q[0] - 1.0) < 1e-14
    np.testing.assert_allclose(ang[0], np.degrees(np.arccos(-1.0 / 3.0)), atol=1e-10)
    # isolated water => q = 0, padded outputs
    lone = np.array([[1.0, 2.0, 3.0], [40.0, 40.0, 40.0]])
    q, idx, dist, ang = oracle.getOrderParamq(lone, lone, np.array([90.0] * 3), full=True)
    assert np.all(q == 0.0) and np.all(idx == -1) and np.all(np.isnan(dist))
    # one neighbour => six 180-degree angles => q = 1 - 3/8*6*(2/3)^2 = 0
    pair = np.array([[1.0, 2.0, 3.0], [2.0, 2.0, 3.0]])
    q, idx, dist, ang = oracle.getOrderParamq(pair, pair, np.array([90.0] * 3), full=True)
    assert np.all(ang == 180.0) and np.all(np.abs(q) < 1e-15) and idx[0, 0] == 1 and idx[1, 0] == 0


def test_willard_density_field(oracle):
    """SURVEY 8f row f-1 (waterlib.f90:1286-1398): the oracle reproduces the reference bit for bit"""
    g = golden("willard_slab700")
    dv, dn = oracle.willarddensityfield(g["pos"], g["gridx"], g["gridy"], g["gridz"], g["box"], float(g["smoothlen"]))
    assert dv.shape == g["densvals"].shape and dv.flags.f_contiguous
    assert np.array_equal(dv, g["densvals"]) and np.array_equal(dn, g["densnorms"], equal_nan=True)
    pv, pn = oracle.willarddensitypoints(g["pos"], g["pts"], g["box"], float(g["smoothlen"]))
    assert np.array_equal(pv, g["pts_densvals"]) and np.array_equal(pn, g["pts_densnorms"], equal_nan=True)


def test_radial_distribution_functions(oracle):
    """SURVEY 8f row f-3 (waterlib.f90:193-231, 316-389): counts, bin rule and the single-precision 4./3. /
    truncated-pi normalisation reproduce the reference bit for bit"""
    g = golden("rdf700")
    pos, box, sol = g["pos"], g["box"], g["sol"]
    assert np.array_equal(oracle.radialdistsame(pos, 0.1, 150, 1.0, box), g["rdf_same"])
    assert np.array_equal(oracle.radialdist(sol, pos, 0.1, 150, 0.0334, box), g["rdf_sol_wat"])
    assert np.array_equal(oracle.radialdistsame(sol, 0.1, 150, 1.0, box), g["rdf_sol_sol"])
    assert np.array_equal(oracle.pairdistancehistogram(sol, pos, 0.25, 60, box), g["hist_sol_wat"])
    assert np.array_equal(oracle.pairdistancehistogram(pos, pos, 0.05, 300, box), g["hist_wat_wat"])
    assert np.array_equal(oracle.radialdistsame(pos, 0.1, 150, 1.0, box * np.array([1, 1, -1.0])), g["rdf_same_nonperiodic_z"])
    # every pair lands in exactly one bin when the range covers the box diagonal
    h = oracle.pairdistancehistogram(pos, pos, 0.5, 100, box)
    assert h.sum() == len(pos) * (len(pos) - 1)


def test_water_orientation(oracle):
    """SURVEY 8f row f-2 (waterlib.f90:954-1044): bit for bit against the reference's outputs"""
    g = golden("orient500")
    ad, ap = oracle.watorient(g["opos"], g["hpos"], g["refvec"], g["box"])
    assert np.array_equal(ad, g["angdip"]) and np.array_equal(ap, g["angplane"])
    assert np.array_equal(oracle.watohvec(g["opos"], g["hpos"], g["box"]), g["normdip"])
    assert np.array_equal(oracle.watorient(g["opos"], g["hpos"], [0.0, 0.0, 1.0], g["box"])[0], g["angdip_z"])


def test_general_hbonds(oracle):
    """SURVEY 8f row f-4 (waterlib.f90:1156-1210): the bond set equals the reference's"""
    g = golden("hbonds600")
    for key, (dc, ac) in (("pairs_3p0_150", (3.0, 150.0)), ("pairs_3p5_120", (3.5, 120.0))):
        b = oracle.generalhbonds(g["acc"], g["don"], g["donh"], g["box"], dc, ac)
        assert b.dtype == np.int32 and b.flags.f_contiguous and b.shape == (600, 1200)
        assert np.array_equal(np.stack(np.where(b == 1), axis=1), g[key])
