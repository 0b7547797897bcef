"""CPU: the oracle (oracle/cpu_ref.py) against the golden vectors captured from the imported
reference (oracle/make_golden.py). This is what pins the oracle; tolerance ~1e-12 relative."""
import numpy as np
import pytest

from oracle import cpu_ref as ref
from radtxfr_amd import synthetic

RT = 2e-13


def close(a, b, rtol=RT, atol=0.0):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_g1_planck(golden):
    g = golden("g1_planck.npz")
    close(ref.planckian(g["X"], g["T32"]), g["L32"])
    assert ref.planckian(g["X"], 296.0).shape == (g["X"].size,)
    close(ref.planckian(g["X"], 296.0), g["Ls"])
    close(ref.planckian(g["Xum"], g["T32"][:5], wavelength=True), g["Lum"])
    close(ref.planckian(g["X"], g["T2d"]), g["L2d"])
    spot = [ref.planckian(500.0, 296)[0], *ref.planckian(1000.0, [250, 300])[0], ref.planckian(10.0, 300, wavelength=True)[0]]
    close(spot, g["spot"])
    # SURVEY 8c anchors
    close(g["spot"], [1.436645756277321e+01, 3.783489547068869, 9.92401679884583, 9.924016798845834e+02], rtol=1e-14)
    close(ref.brightnessTemperature(g["X"], g["L32"]), g["BT"], rtol=1e-12)
    close(ref.brightnessTemperature(g["Xum"], g["Lum"], wavelength=True), g["BTum"], rtol=1e-12)
    close(ref.BT2L(g["X"], np.tile(g["T32"][None, :8], (g["X"].size, 1))), g["L_bt2l"])


def test_g2_cpf_and_voigt(golden):
    g = golden("g2_cpf_voigt.npz")
    a, L = ref.weideman_coeffs(24)
    close(a, g["w24"], rtol=0, atol=1e-18)
    assert L == float(g["L24"])
    wr, wi = ref.hum1_wei(g["x"], g["y"])
    close(wr, g["wr"], rtol=1e-12, atol=1e-300)
    close(wi, g["wi"], rtol=1e-12, atol=1e-300)
    close(ref.PROFILE_VOIGT(1000, 0.0009, 0.07, g["sg"])[0], g["pv"], rtol=1e-12)
    close(ref.PROFILE_VOIGT(2350.0123, 0.0022, 0.004, g["sg2"])[0], g["pv2"], rtol=1e-12)
    close(g["pv"], [2.2173082542317951e-02, 1.4955143584005925e+00, 4.5467419204585502e+00,
                    3.0111229734373595e+00, 1.8181866207165288e-03], rtol=1e-13)


def test_g3_tips(golden):
    g = golden("g3_tips.npz")
    for r, (m, i) in enumerate(g["mi"].tolist()):
        q = [ref.PYTIPS(m, i, float(t)) for t in g["T"]]
        close(q, g["Q"][r], rtol=1e-14)
    close(ref.PYTIPS(1, 1, 296.0), 174.638204378906, rtol=1e-13)
    close(ref.PYTIPS(2, 1, 287.87), 277.004350581581, rtol=1e-13)
    close(ref.PYTIPS(3, 1, 216.7), 2107.312392808672, rtol=1e-13)
    s = ref.EnvironmentDependency_Intensity(1e-21, 250., 296., ref.PYTIPS(1, 1, 250.), ref.PYTIPS(1, 1, 296.), 500., 1000.)
    close(s, float(g["S_spot"]), rtol=1e-14)
    with pytest.raises(Exception):
        ref.PYTIPS(1, 1, 69.9)
    with pytest.raises(Exception):
        ref.PYTIPS(1, 1, 3000.1)


def test_g4_voigt_xsec(golden):
    g = golden("g4_voigt_xsec.npz")
    tbl = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    grid = np.linspace(float(g["grid_lo"]), float(g["grid_hi"]), int(g["grid_n"]))
    for tag in ("l01", "l32"):
        om, xs = ref.absorptionCoefficient_Voigt(tbl, T=float(g["T_" + tag]), p=float(g["p_" + tag]), OmegaGrid=grid)
        assert np.array_equal(om, grid)
        close(xs, g["xs_" + tag], rtol=1e-11, atol=1e-40)
    _, xs = ref.absorptionCoefficient_Voigt(tbl, Components=[(1, 1), (2, 1, 0.5)], T=250.0, p=0.4,
                                            OmegaGrid=grid[20000:30000], HITRAN_units=False, GammaL="gamma_self",
                                            OmegaWing=2.0, OmegaWingHW=20.0)
    close(xs, g["xs_opt"], rtol=1e-11, atol=1e-30)
    _, xs = ref.absorptionCoefficient_Voigt(tbl, T=230.0, p=0.05, OmegaGrid=grid[40000:46000],
                                            Diluent={"air": 0.7, "self": 0.3})
    close(xs, g["xs_dil"], rtol=1e-11, atol=1e-40)


def test_g5_tud_windows(golden):
    g = golden("g5_tud_windows.npz")
    full = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    a = synthetic.c3_atmosphere(32)
    pad = float(g["pad"])
    for w in range(3):
        lo, hi = float(g[f"w{w}_lo"]), float(g[f"w{w}_hi"])
        sub = synthetic.subset_table(full, lo - pad, hi + pad)
        X, tau, Lu, Ld, OD = ref.compute_TUD(sub, lo, hi, 0.001, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"],
                                             a["MFs_ID"], return_layers=True)
        assert np.array_equal(X, g[f"w{w}_X"])
        close(OD, g[f"w{w}_OD"], rtol=1e-11, atol=1e-30)
        close(tau, g[f"w{w}_tau"], rtol=1e-11)
        close(Lu, g[f"w{w}_Lu"], rtol=1e-11)
        close(Ld, g[f"w{w}_Ld"], rtol=1e-11)
        if w == 1:
            t2, u2, d2 = ref.tud_from_od(X, OD, a["Ts"], a["Zs"], Altitudes=[1.0, 4.05, 9.0], theta_r=0.6, N_angle=7)
            assert t2.shape == (X.size, 3)
            close(t2, g["w1_tau_alt"], rtol=1e-11)
            close(u2, g["w1_Lu_alt"], rtol=1e-11)
            close(d2, g["w1_Ld_alt"], rtol=1e-11)
            t3, u3, d3 = ref.tud_from_od(X, OD, a["Ts"], a["Zs"], Altitudes=[9.0], returnOD=True)
            close(t3, g["w1_tau_rod"], rtol=1e-11)
            close(u3, g["w1_Lu_rod"], rtol=1e-11)
            close(d3, g["w1_Ld_rod"], rtol=1e-11)


def test_g8_tud_thin(golden):
    g = golden("g8_tud_thin.npz")
    full = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    A = synthetic.load_standard_atmosphere()
    for tag in "ab":
        lo, hi, nl = float(g[tag + "_lo"]), float(g[tag + "_hi"]), int(g[tag + "_nlay"])
        sub = synthetic.subset_table(full, lo - float(g["pad"]), hi + float(g["pad"]))
        a = A[:nl]
        X, tau, Lu, Ld = ref.compute_TUD(sub, lo, hi, 0.001, a[:, 1], a[:, 5], a[:, 4], a[:, 3], a[:, 6:8] * 1e6 * float(g[tag + "_scale"]),
                                         np.array([1, 2]), theta_r=float(g["theta_r"]))
        close(tau, g[tag + "_tau"], rtol=1e-11)
        close(Lu, g[tag + "_Lu"], rtol=1e-11)
        close(Ld, g[tag + "_Ld"], rtol=1e-11)


def test_g6_apparent_radiance(golden):
    g = golden("g6_apparent_radiance.npz")
    L0 = ref.compute_LWIR_apparent_radiance(g["X"], g["emis"], g["Ts"], g["tau"], g["La"], g["Ld"])
    close(L0, g["L0"])
    L1, Ls1 = ref.compute_LWIR_apparent_radiance(g["X"], g["emis"], g["Ts"], g["tau"], g["La"], g["Ld"], dT=g["dT"], return_Ls=True)
    assert L1.shape == (128, 9, 3, 9)
    close(L1, g["L1"])
    close(Ls1, g["Ls1"])


def test_g7_ils(golden):
    g = golden("g7_ils.npz")
    X = np.linspace(float(g["X_lo"]), float(g["X_hi"]), int(g["X_n"]))
    Y2 = g["Y2"]
    xo, yo = ref.ILS_MAKO(X, Y2[:, 0])
    close(xo, g["xo1"], rtol=0)
    close(yo, g["yo1"], rtol=1e-12)
    xo, yo = ref.ILS_MAKO(X, Y2)
    close(xo, g["xo2"], rtol=0)
    close(yo, g["yo2"], rtol=1e-12)
    xo, yo = ref.ILS_MAKO(X, Y2, resFactor=2)
    assert xo.size == g["xo3"].size
    close(xo, g["xo3"], rtol=0)
    close(yo, g["yo3"], rtol=1e-12)
    yo = ref.ILS_MAKO(X, Y2, returnX=False, fwhm_sf=1.3, shift=0.4, scale=1.0005)
    close(yo, g["yo4"], rtol=1e-12)
    xg, yg = ref.ILS_MAKO_gauss(X, Y2[:, 0])
    close(xg, g["xg1"], rtol=0)
    close(yg, g["yg1"], rtol=1e-12)
    xg, yg = ref.ILS_MAKO_gauss(X, Y2)
    close(yg, g["yg2"], rtol=1e-12)


def test_g9_smooth_and_reduce_resolution(golden):
    """Oracle restatement of radiative_transfer.smooth / reduceResolution against the reference run (np.int restored)."""
    g = golden("g9_reduce.npz")
    assert np.allclose(ref.smooth(g["Y1"], 11), g["sm11"], rtol=1e-13, atol=0)
    assert np.allclose(ref.smooth(g["Y1"], 50, "hamming"), g["sm50"], rtol=1e-13, atol=0)
    assert np.allclose(ref.smooth(g["Y1"], 7, "flat"), g["smflat"], rtol=1e-13, atol=0)
    Xo, Yo = ref.reduceResolution(g["Xf"], g["Y1"], 0.05)
    assert Xo.shape == g["Xo"].shape and np.allclose(Xo, g["Xo"], rtol=1e-14, atol=0)
    assert np.allclose(Yo, g["Yo1"], rtol=1e-11, atol=0)
    assert np.allclose(ref.reduceResolution(g["Xf"], g["Y2"], 0.05, X_out=g["Xo"]), g["Yo2"], rtol=1e-11, atol=0)
    Xo8, Yo8 = ref.reduceResolution(g["Xf"], g["Y1"], 0.02, N=8, window="blackman")
    assert Xo8.shape == g["Xo8"].shape and np.allclose(Yo8, g["Yo8"], rtol=1e-11, atol=0)


def test_g10_lorentz_doppler(golden):
    """Oracle restatements of absorptionCoefficient_Lorentz / _Doppler against the reference run."""
    g = golden("g10_lorentz_doppler.npz")
    tbl = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    gl = np.linspace(float(g["gl_lo"]), float(g["gl_hi"]), int(g["gl_n"]))
    gd = np.linspace(float(g["gd_lo"]), float(g["gd_hi"]), int(g["gd_n"]))
    for tag in ("l01", "l32"):
        _, xs = ref.absorptionCoefficient_Lorentz(tbl, T=float(g["T_" + tag]), p=float(g["p_" + tag]), OmegaGrid=gl)
        close(xs, g["lor_" + tag], rtol=1e-11, atol=1e-40)
    _, xs = ref.absorptionCoefficient_Lorentz(tbl, Components=[(1, 1), (2, 1, 0.5)], T=250.0, p=0.4, OmegaGrid=gl[5000:12000],
                                              HITRAN_units=False, OmegaWing=1.0, OmegaWingHW=20.0,
                                              Diluent={"air": 0.7, "self": 0.3})
    close(xs, g["lor_opt"], rtol=1e-11, atol=1e-30)
    for tag, (Tk, pk) in (("a", (296.0, 1.0)), ("b", (220.0, 0.05))):
        _, xs = ref.absorptionCoefficient_Doppler(tbl, T=Tk, p=pk, OmegaGrid=gd)
        close(xs, g["dop_" + tag], rtol=1e-10, atol=1e-40)
    _, xs = ref.absorptionCoefficient_Doppler(tbl, LineShift=False, T=296.0, p=1.0, OmegaGrid=gd, HITRAN_units=False, OmegaWing=0.05)
    close(xs, g["dop_noshift"], rtol=1e-10, atol=1e-30)


def _g11_table(g):
    tbl = dict(synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"])))
    tbl["SD_air"], tbl["SD_self"] = g["SD_air"], g["SD_self"]
    return tbl, np.linspace(float(g["g_lo"]), float(g["g_hi"]), int(g["g_n"]))


def test_g11_sdvoigt(golden):
    """Oracle restatement of PROFILE_SDVOIGT / pcqsdhc (anuVC = eta = 0) against the reference run."""
    g = golden("g11_sdvoigt.npz")
    tbl, grid = _g11_table(g)
    _, xs = ref.absorptionCoefficient_SDVoigt(tbl, T=250.0, p=0.3, OmegaGrid=grid)
    close(xs, g["xs_a"], rtol=1e-11, atol=1e-40)
    _, xs = ref.absorptionCoefficient_SDVoigt(tbl, T=296.0, p=1.0, OmegaGrid=grid, Diluent={"air": 0.6, "self": 0.4})
    close(xs, g["xs_b"], rtol=1e-11, atol=1e-40)
    _, xs = ref.absorptionCoefficient_SDVoigt(tbl, T=220.0, p=0.01, OmegaGrid=grid)
    close(xs, g["xs_c"], rtol=1e-11, atol=1e-40)
    _, xs = ref.absorptionCoefficient_SDVoigt(tbl, T=300.0, p=0.8, OmegaGrid=grid, HITRAN_units=False, OmegaWing=0.5, OmegaWingHW=20.0,
                                              Components=[(1, 1), (2, 1, 0.5)])
    close(xs, g["xs_d"], rtol=1e-11, atol=1e-30)
    _, xv = ref.absorptionCoefficient_Voigt(tbl, T=250.0, p=0.3, OmegaGrid=grid)
    assert np.max(np.abs(xv - g["xs_a"])) / np.max(xv) > 5e-3  # the speed dependence is not a rounding-level effect here


def _g12_tables(g):
    """Numeric line-table columns exactly as the reference parsed them (tests/golden/g12_par_tables.npz)."""
    cols = ("molec_id", "local_iso_id", "nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air")
    a = {k: g["g12a_" + k] for k in cols}
    b = {k: g["g12b_" + k] for k in cols + ("n_self", "deltap_air", "delta_self", "deltap_self")}
    return a, b


def test_g12_voigt_on_parsed_par_tables(golden):
    """The reference's Voigt line-sum on the rows ITS parser kept (air only; air + self with n_self, deltap_air,
    delta_self, deltap_self): pins the oracle's optional-column handling (misc/hapi.py:11090-11128)."""
    g = golden("g12_par_tables.npz")
    a, b = _g12_tables(g)
    grid = np.linspace(float(g["grid_lo"]), float(g["grid_hi"]), int(g["grid_n"]))
    _, xs = ref.absorptionCoefficient_Voigt(a, T=float(g["T"]), p=float(g["p"]), OmegaGrid=grid)
    close(xs, g["g12a_xs"], rtol=1e-11, atol=1e-40)
    _, xs = ref.absorptionCoefficient_Voigt(b, T=float(g["T"]), p=float(g["p"]), OmegaGrid=grid, Diluent={"air": 0.6, "self": 0.4})
    close(xs, g["g12b_xs"], rtol=1e-11, atol=1e-40)
    assert np.any(b["n_self"] == 0.0) and np.any(b["deltap_self"] != 0.0)


def test_g13_tud_vector_theta(golden):
    """Vector theta_r (radiative_transfer.py:313, 346-365): (nX, nZ, nMu) and the (nX, nMu) squeeze."""
    g = golden("g13_tud_slants.npz")
    full = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    lo, hi = float(g["lo"]), float(g["hi"])
    sub = synthetic.subset_table(full, lo - float(g["pad"]), hi + float(g["pad"]))
    a = synthetic.c3_atmosphere(32)
    mf = a["MFs_VAL"] * float(g["mf_scale"])
    X, tau, Lu, Ld = ref.compute_TUD(sub, lo, hi, 0.001, a["Zs"], a["Ts"], a["Ps"], a["PLs"], mf, a["MFs_ID"],
                                     Altitudes=g["alt22"], theta_r=g["th22"])
    assert tau.shape == (X.size, 2, 2)
    close(tau, g["tau22"], rtol=1e-11)
    close(Lu, g["Lu22"], rtol=1e-11)
    close(Ld, g["Ld22"], rtol=1e-11)
    X, tau, Lu, Ld = ref.compute_TUD(sub, lo, hi, 0.001, a["Zs"], a["Ts"], a["Ps"], a["PLs"], mf, a["MFs_ID"], theta_r=g["th13"])
    assert tau.shape == (X.size, 3)
    close(tau, g["tau13"], rtol=1e-11)
    close(Lu, g["Lu13"], rtol=1e-11)
    close(Ld, g["Ld13"], rtol=1e-11)
