"""GPU parity tests proper: the HIP path (through the C ABI via the reference-shaped shims) against
(a) the golden vectors captured from the imported reference and (b) the CPU oracle on seeded inputs.

Tolerances (SURVEY.md 8d, fp64 reference -> fp32 engine):
    radiances / cross sections / OD : max |x-ref| / max(|ref|, 1e-3 max|ref|) <= 1e-5
    transmittance                   : |dtau| <= 2e-6 absolute
    fp64 entry points (planckian)   : 1e-12 relative
"""
import os

import numpy as np
import pytest

from conftest import rel_err
from oracle import cpu_ref as ref
from radtxfr_amd import synthetic

pytestmark = pytest.mark.gpu

TOL_L = 1e-5
TOL_TAU = 2e-6


@pytest.fixture(scope="module")
def rt():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test without a GPU"
    from radtxfr_amd import _lib
    _lib.load()
    from radtxfr_amd import radiative_transfer
    return radiative_transfer


@pytest.fixture(scope="module")
def hapi(rt):
    from radtxfr_amd import hapi as h
    return h


def test_native_library_loaded(rt):
    import os
    from radtxfr_amd import _lib
    with open("/proc/self/maps") as f:
        assert any(os.path.basename(_lib.LIB_PATH) in ln for ln in f), "libradtxfr_hip.so is not mapped"


# ------------------------------------------------------------------------------------ G1 planckian
def test_g1_planckian(rt, golden):
    g = golden("g1_planck.npz")
    L = rt.planckian(g["X"], g["T32"])
    assert L.shape == g["L32"].shape and L.dtype == np.float64
    np.testing.assert_allclose(L, g["L32"], rtol=1e-12)
    Ls = rt.planckian(g["X"], 296.0)
    assert Ls.shape == (g["X"].size,)
    np.testing.assert_allclose(Ls, g["Ls"], rtol=1e-12)
    np.testing.assert_allclose(rt.planckian(g["Xum"], g["T32"][:5], wavelength=True), g["Lum"], rtol=1e-12)
    L2 = rt.planckian(g["X"], g["T2d"])
    assert L2.shape == (g["X"].size, 2, 3)
    np.testing.assert_allclose(L2, g["L2d"], rtol=1e-12)
    np.testing.assert_allclose(rt.planckian(500.0, 296)[0], 1.436645756277321e+01, rtol=1e-12)
    # wavelength heuristic mean(X) < 50 (:836)
    np.testing.assert_allclose(rt.planckian(np.array([10.0]), 300.0)[0], 9.924016798845834e+02, rtol=1e-12)


def test_g1_brightness_temperature_and_bt2l(rt, golden):
    g = golden("g1_planck.npz")
    BT = rt.brightnessTemperature(g["X"], g["L32"])
    assert BT.shape == g["BT"].shape
    np.testing.assert_allclose(BT, g["BT"], rtol=1e-12)
    np.testing.assert_allclose(rt.brightnessTemperature(g["Xum"], g["Lum"], wavelength=True), g["BTum"], rtol=1e-12)
    T8 = np.tile(g["T32"][None, :8], (g["X"].size, 1))
    np.testing.assert_allclose(rt.BT2L(g["X"], T8), g["L_bt2l"], rtol=1e-12)
    # round trip, 1-D quirk (a vector comes back (nX,1)), spectral_dim, bad-value masking (:922-923, :1004-1005)
    L1 = rt.planckian(g["X"], 300.0)
    T1 = rt.brightnessTemperature(g["X"], L1)
    assert T1.shape == (g["X"].size, 1)
    np.testing.assert_allclose(T1[:, 0], 300.0, rtol=1e-12)
    np.testing.assert_allclose(rt.brightnessTemperature(g["X"], g["L32"].T, spectral_dim=1), g["BT"].T, rtol=1e-12)
    Lb = g["L32"].copy()
    Lb[3, 2], Lb[5, 0], Lb[7, 1] = -1.0, 0.0, np.inf
    Tb = rt.brightnessTemperature(g["X"], Lb, bad_value=-99.0)
    Tr = ref.brightnessTemperature(g["X"], Lb, bad_value=-99.0)
    assert Tb[3, 2] == Tb[5, 0] == Tb[7, 1] == -99.0
    np.testing.assert_allclose(Tb, Tr, rtol=1e-12)
    Tn = T8.copy()
    Tn[2, 2] = -5.0
    np.testing.assert_allclose(rt.BT2L(g["X"], Tn, bad_value=0.0), ref.BT2L(g["X"], Tn, bad_value=0.0), rtol=1e-12)


# --------------------------------------------------------------------- G4 absorptionCoefficient_Voigt
def _g4_table(hapi, g):
    tbl = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    hapi.storage2cache_from_columns("g4", tbl)
    return np.linspace(float(g["grid_lo"]), float(g["grid_hi"]), int(g["grid_n"]))


@pytest.mark.parametrize("tag", ["l01", "l32"])
def test_g4_voigt_xsec_golden(hapi, golden, tag):
    g = golden("g4_voigt_xsec.npz")
    grid = _g4_table(hapi, g)
    om, xs = hapi.absorptionCoefficient_Voigt(SourceTables="g4", Environment={"T": float(g["T_" + tag]), "p": float(g["p_" + tag])},
                                              OmegaGrid=grid, HITRAN_units=True)
    assert np.array_equal(om, grid) and xs.dtype == np.float64
    assert rel_err(xs, g["xs_" + tag]) <= TOL_L


def test_g4_voigt_options_golden(hapi, golden):
    g = golden("g4_voigt_xsec.npz")
    grid = _g4_table(hapi, g)
    _, xs = hapi.absorptionCoefficient_Voigt(Components=[(1, 1), (2, 1, 0.5)], SourceTables="g4", Environment={"T": 250.0, "p": 0.4},
                                             OmegaGrid=grid[20000:30000], HITRAN_units=False, GammaL="gamma_self",
                                             OmegaWing=2.0, OmegaWingHW=20.0)
    assert rel_err(xs, g["xs_opt"]) <= TOL_L
    _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="g4", Environment={"T": 230.0, "p": 0.05}, OmegaGrid=grid[40000:46000],
                                             Diluent={"air": 0.7, "self": 0.3})
    assert rel_err(xs, g["xs_dil"]) <= TOL_L


def test_voigt_window_edges_exact(hapi):
    """The hard wing cutoff must land on the same grid points as bisect() (quirk 8): compare the
    support (non-zero set) of single-line cross sections with the oracle's, point for point."""
    tbl = synthetic.synth_line_table(99, 40, 995.0, 1005.0)
    grid = np.linspace(990.0, 1010.0, 20000)
    for r in range(0, 40, 3):
        one = {k: v[r:r + 1] for k, v in tbl.items()}
        hapi.storage2cache_from_columns("one", one)
        for T, p in ((296.0, 1.0), (220.0, 0.02)):
            _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="one", Environment={"T": T, "p": p}, OmegaGrid=grid)
            _, xr = ref.absorptionCoefficient_Voigt(one, T=T, p=p, OmegaGrid=grid)
            assert np.array_equal(xs != 0, xr != 0)
            assert rel_err(xs, xr) <= TOL_L


def _clustered():
    tbl = synthetic.synth_clustered_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    h, edges = np.histogram(tbl["nu"], bins=np.arange(475.0, 6026.0, 0.5))
    return tbl, edges, h


def test_clustered_table_band_head_vs_oracle(hapi):
    """A line list shaped like HITRAN rather than uniform noise: a band head with 3000 lines inside
    0.5 cm^-1 -- a quarter of them distinct centres, the rest exact repeats -- strengths spanning 1e-30 ... 1e-19 within one
    tile, a P/R comb on both sides: thousands of candidates per tile, many identical nearest-grid indices. Against the
    oracle at 0.001 and 0.0005 cm^-1, at the surface and at 0.01 atm (Doppler-dominated: the fp64 band lanes); plus an
    isolated line in an otherwise empty span (most tiles of its window hold nothing else)."""
    tbl, edges, h = _clustered()
    head = float(edges[int(np.argmax(h))])
    assert h.max() >= 2500
    for step in (0.001, 0.0005):
        grid = np.linspace(head - 1.0, head + 1.5, int(round(2.5 / step)) + 1)
        sub = synthetic.subset_table(tbl, grid[0] - 12.0, grid[-1] + 12.0)
        assert sub["nu"].size > 3000 and np.unique(sub["nu"]).size < sub["nu"].size - 1000
        hapi.storage2cache_from_columns("clu", sub)
        for T, p in ((287.9, 0.994), (220.0, 0.01)):
            _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="clu", Environment={"T": T, "p": p}, OmegaGrid=grid)
            _, xr = ref.absorptionCoefficient_Voigt(sub, T=T, p=p, OmegaGrid=grid)
            assert xr.max() > 100 * np.median(xr) or xr.min() > 0
            assert rel_err(xs, xr) <= TOL_L, (step, T, p, rel_err(xs, xr))
    # an isolated line: the emptiest 20 cm^-1 around a line centre
    nu = tbl["nu"]
    gap = np.minimum(np.diff(nu)[1:], np.diff(nu)[:-1])  # distance to the nearer neighbour, for lines 1 .. n-2
    r = int(np.argmax(gap)) + 1
    assert gap[r - 1] > 3.0
    grid = np.linspace(nu[r] - 6.0, nu[r] + 6.0, 12001)
    sub = synthetic.subset_table(tbl, grid[0] - 12.0, grid[-1] + 12.0)
    hapi.storage2cache_from_columns("clu", sub)
    _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="clu", Environment={"T": 287.9, "p": 0.994}, OmegaGrid=grid)
    _, xr = ref.absorptionCoefficient_Voigt(sub, T=287.9, p=0.994, OmegaGrid=grid)
    assert np.array_equal(xs != 0, xr != 0) and (xr == 0).sum() > 100 and rel_err(xs, xr) <= TOL_L
    hapi.LOCAL_TABLE_CACHE.pop("clu")


def test_clustered_table_full_grid_properties():
    """The clustered table on the FULL C3 grid x 32 layers: optical depth of every layer against the oracle on windows at a
    band head, on its comb and in an empty stretch; the same bits from run to run; and a tile-aligned shard computed from
    only the lines in reach of it (what a rank of N holds) bit-identical to the same slice of the full-table, full-grid
    run -- the candidate ranges are trimmed to the lines that reach each tile, so neither the table subset nor a hot
    tile's thousands of candidates change the order of the sums."""
    import torch
    from radtxfr_amd import _lib, engine
    tbl, edges, h = _clustered()
    a = synthetic.c3_atmosphere(32)
    lines = engine.LineTable(tbl)
    grid = engine.Grid(500.0, 6000.0, 5500000)
    OD = engine.optical_depths(lines, grid, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(OD).all()) and float(OD.min()) >= 0.0
    # the band heads put > 768 candidates on some tiles: those are cut into parts (rtx_prep_split_bound > 0); the uniform
    # benchmark table has no such tile and never launches the extra kernels
    lib = _lib.load()
    assert lib.rtx_prep_split_bound(lines.plan(32, grid.n)._h) > 100
    OD_b = engine.optical_depths(lines, grid, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert torch.equal(OD, OD_b), "not bit-reproducible from run to run"
    # rtx_voigt_sum twice after ONE prologue (the C ABI allows it): the hot-tile work list is rebuilt, not appended to
    import ctypes as C
    plan = lines.plan(32, grid.n)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    OD_c = torch.empty_like(OD)
    _lib.check(lib.rtx_voigt_sum(plan._h, grid.byref(), 32, C.c_void_p(OD_c.data_ptr()), None, grid.n, st))
    assert torch.equal(OD, OD_c), "a second sum after the same prologue differs"
    del OD_c
    uni = engine.LineTable(synthetic.subset_table(synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0), 900.0, 1100.0))
    g_u = engine.Grid(500.0, 6000.0, 5500000).shard(450 * 1024, 64 * 1024)
    engine.optical_depths(uni, g_u, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert lib.rtx_prep_split_bound(uni.plan(32, g_u.n)._h) == 0
    uni.close()
    del OD_b
    X = grid.axis()
    inside = (edges[:-1] > 520.0) & (edges[:-1] < 5980.0)
    head = float(edges[:-1][inside][int(np.argmax(h[inside]))])
    empty = float(edges[:-1][inside][int(np.argmin(h[inside]))])
    for x0, n in ((head - 0.2, 900), (head + 6.0, 900), (empty, 900)):
        i0 = int((x0 - 500.0) / grid.step)
        Xw = X[i0:i0 + n]
        sub = synthetic.subset_table(tbl, Xw[0] - 12.0, Xw[-1] + 12.0)
        ODr = np.stack([ref.layer_od(sub, Xw, a["Ts"][k], a["Ps"][k], a["PLs"][k], a["MFs_VAL"][k], a["MFs_ID"]) for k in range(32)], 1)
        got = OD[:, i0:i0 + n].T.double().cpu().numpy()
        if ODr.max() == 0.0:
            assert got.max() == 0.0
        else:
            assert rel_err(got, ODr) <= TOL_L, (x0, rel_err(got, ODr))
    # a rank's view: tile-aligned shard around the head, only the lines in reach
    tp = int(_lib.load().rtx_voigt_tile_points())
    t0 = int((head - 500.0) / grid.step) // tp - 3
    sh = grid.shard(t0 * tp, 9 * tp)
    reach = engine.max_wing_cm(tbl, a["Ts"], a["Ps"] / 101325.0) + grid.step
    sub = synthetic.subset_table(tbl, sh.x_at(0) - reach, sh.x_at(sh.n - 1) + reach)
    assert 3000 < sub["nu"].size < 20000
    lines_s = engine.LineTable(sub)
    OD_s = engine.optical_depths(lines_s, sh, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert torch.equal(OD_s, OD[:, sh.offset:sh.offset + sh.n]), "a rank's shard (subset table) differs from the full run"
    lines_s.close()
    lines.close()


def test_voigt_doppler_regime_vs_oracle(hapi):
    """Low pressure / high wavenumber: y << 1, most of the window inside the Weideman region."""
    tbl = synthetic.synth_line_table(5, 300, 4990.0, 5010.0)
    hapi.storage2cache_from_columns("dop", tbl)
    grid = np.linspace(4995.0, 5005.0, 50000)
    for T, p in ((250.0, 1e-3), (296.0, 0.05), (200.0, 1e-5)):
        _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="dop", Environment={"T": T, "p": p}, OmegaGrid=grid)
        _, xr = ref.absorptionCoefficient_Voigt(tbl, T=T, p=p, OmegaGrid=grid)
        assert rel_err(xs, xr) <= TOL_L, (T, p)


def test_voigt_edge_cases(hapi):
    tbl = synthetic.synth_line_table(3, 50, 900.0, 1100.0)
    hapi.storage2cache_from_columns("e", tbl)
    # grid entirely away from every line: exact zeros
    g0 = np.linspace(3000.0, 3001.0, 777)
    _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="e", OmegaGrid=g0)
    assert xs.shape == (777,) and np.all(xs == 0)
    # ragged size (not a multiple of any tile), lines left and right of the grid, tiny grid
    for n in (2, 63, 257, 2049, 5001):
        g1 = np.linspace(990.0, 1010.0, n)
        _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="e", OmegaGrid=g1)
        _, xr = ref.absorptionCoefficient_Voigt(tbl, OmegaGrid=g1)
        assert rel_err(xs, xr) <= TOL_L, n
    # Components filter that selects nothing -> zeros; unknown table / component -> exceptions like hapi
    _, xs = hapi.absorptionCoefficient_Voigt(Components=[(5, 1)], SourceTables="e", OmegaGrid=g1)
    assert np.all(xs == 0)
    with pytest.raises(Exception):
        hapi.absorptionCoefficient_Voigt(SourceTables="nope", OmegaGrid=g1)
    with pytest.raises(Exception):
        hapi.absorptionCoefficient_Voigt(Components=[(99, 1)], SourceTables="e", OmegaGrid=g1)
    with pytest.raises(Exception):  # TIPS range (:9571)
        hapi.absorptionCoefficient_Voigt(SourceTables="e", Environment={"T": 50.0, "p": 1.0}, OmegaGrid=g1)
    # OmegaRange/OmegaStep path (arange_) and IntensityThreshold
    om, xs = hapi.absorptionCoefficient_Voigt(SourceTables="e", OmegaRange=(950.0, 1050.0), OmegaStep=0.01, IntensityThreshold=1e-23)
    _, xr = ref.absorptionCoefficient_Voigt(tbl, OmegaGrid=om, IntensityThreshold=1e-23)
    _, xall = ref.absorptionCoefficient_Voigt(tbl, OmegaGrid=om)
    assert om.size == 10001 and np.any(xr != 0) and np.any(xall != xr)  # the threshold really drops some lines
    assert rel_err(xs, xr) <= TOL_L


# ----------------------------------------------------------------------- G5 compute_OD / compute_TUD
def _g5(golden):
    g = golden("g5_tud_windows.npz")
    full = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    return g, full, synthetic.c3_atmosphere(32), float(g["pad"])


@pytest.mark.parametrize("w", [0, 1, 2])
def test_g5_compute_tud_golden(rt, golden, w):
    g, full, a, pad = _g5(golden)
    lo, hi = float(g[f"w{w}_lo"]), float(g[f"w{w}_hi"])
    sub = synthetic.subset_table(full, lo - pad, hi + pad)
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, Altitudes=np.asarray([500]), theta_r=0, N_angle=30, **a)
    assert np.array_equal(X, g[f"w{w}_X"])
    assert tau.shape == Lu.shape == Ld.shape == X.shape
    assert np.max(np.abs(tau - g[f"w{w}_tau"])) <= TOL_TAU
    assert rel_err(Lu, g[f"w{w}_Lu"]) <= TOL_L
    assert rel_err(Ld, g[f"w{w}_Ld"]) <= TOL_L
    # per-layer optical depth through compute_OD (layer 1 and layer 32)
    for k in (0, 31):
        Xo, od = rt.compute_OD(lo, hi, DVOUT=0.001, line_table=sub, T=a["Ts"][k], P=a["Ps"][k], PL=a["PLs"][k],
                               MF_VAL=a["MFs_VAL"][k], MF_ID=a["MFs_ID"])
        assert np.array_equal(Xo, X)
        assert rel_err(od, g[f"w{w}_OD"][:, k]) <= TOL_L


def test_g5_compute_tud_quirks_golden(rt, golden):
    g, full, a, pad = _g5(golden)
    lo, hi = float(g["w1_lo"]), float(g["w1_hi"])
    sub = synthetic.subset_table(full, lo - pad, hi + pad)
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, theta_r=0.6, Altitudes=np.asarray([1.0, 4.05, 9.0]),
                                    N_angle=7, **a)
    assert tau.shape == (X.size, 3) and Lu.shape == (X.size, 3)
    assert np.max(np.abs(tau - g["w1_tau_alt"])) <= TOL_TAU
    assert rel_err(Lu, g["w1_Lu_alt"]) <= TOL_L
    assert rel_err(Ld, g["w1_Ld_alt"]) <= TOL_L
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, Altitudes=np.asarray([9.0]), returnOD=True, **a)
    assert rel_err(tau, g["w1_tau_rod"]) <= TOL_L  # sum(OD*mu) in the tau slot
    assert rel_err(Lu, g["w1_Lu_rod"]) <= TOL_L
    assert rel_err(Ld, g["w1_Ld_rod"]) <= TOL_L


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g8_compute_tud_thin_golden(rt, golden, tag):
    """Optically thin cases (tau spans (0,1); 66 layers with Doppler-dominated lines in case b): the
    SURVEY 8d table itself is nearly opaque, which would leave tau and the layer weighting untested."""
    g = golden("g8_tud_thin.npz")
    full = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    lo, hi, nl = float(g[tag + "_lo"]), float(g[tag + "_hi"]), int(g[tag + "_nlay"])
    sub = synthetic.subset_table(full, lo - float(g["pad"]), hi + float(g["pad"]))
    a = synthetic.load_standard_atmosphere()[:nl]
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, Zs=a[:, 1], Ts=a[:, 5], Ps=a[:, 4], PLs=a[:, 3],
                                    MFs_VAL=a[:, 6:8] * 1e6 * float(g[tag + "_scale"]), MFs_ID=np.array([1, 2]),
                                    theta_r=float(g["theta_r"]), Altitudes=np.asarray([500]))
    assert g[tag + "_tau"].max() - g[tag + "_tau"].min() > 0.3
    assert np.max(np.abs(tau - g[tag + "_tau"])) <= TOL_TAU
    assert rel_err(Lu, g[tag + "_Lu"]) <= TOL_L
    assert rel_err(Ld, g[tag + "_Ld"]) <= TOL_L


def test_compute_tud_options_do_not_leak(rt, golden):
    """Divergence from quirk 2, on purpose: kwargs must not persist in the module-level options."""
    before = rt.options["DVOUT"]
    g, full, a, pad = _g5(golden)
    sub = synthetic.subset_table(full, 488.0, 514.0)
    rt.compute_TUD(500.0, 500.5, DVOUT=0.01, line_table=sub, **a)
    assert rt.options["DVOUT"] == before and rt.options["line_table"] is None
    with pytest.raises(Exception):
        rt.compute_TUD(500.0, 500.5, DVOUT=0.01)  # no line table configured


def test_tud_66_layers_vs_oracle(rt):
    """All 66 rows of the standard atmosphere (0-100 km: Doppler-dominated upper layers, NL=96 kernel)."""
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi = 2300.0, 2300.8
    sub = synthetic.subset_table(full, lo - 12.0, hi + 12.0)
    A = synthetic.load_standard_atmosphere()
    a = dict(Zs=A[:, 1], Ts=A[:, 5], Ps=A[:, 4], PLs=A[:, 3], MFs_VAL=A[:, 6:8] * 1e6, MFs_ID=np.array([1, 2]))
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.0005, line_table=sub, **a)
    Xr, tau_r, Lu_r, Ld_r = ref.compute_TUD(sub, lo, hi, 0.0005, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert np.array_equal(X, Xr)
    assert np.max(np.abs(tau - tau_r)) <= TOL_TAU
    assert rel_err(Lu, Lu_r) <= TOL_L
    assert rel_err(Ld, Ld_r) <= TOL_L


# ------------------------------------------------------------------------------ G6 apparent radiance
def test_g6_apparent_radiance_golden(rt, golden):
    g = golden("g6_apparent_radiance.npz")
    L0 = rt.compute_LWIR_apparent_radiance(g["X"], g["emis"], g["Ts"], g["tau"], g["La"], g["Ld"])
    assert L0.shape == g["L0"].shape
    assert rel_err(L0, g["L0"]) <= TOL_L
    L1, Ls1 = rt.compute_LWIR_apparent_radiance(g["X"], g["emis"], g["Ts"], g["tau"], g["La"], g["Ld"], dT=g["dT"], return_Ls=True)
    assert L1.shape == g["L1"].shape == (128, 9, 3, 9)
    assert rel_err(L1, g["L1"]) <= TOL_L
    assert rel_err(Ls1, g["Ls1"]) <= TOL_L
    # the reference caller's own usage: float32 inputs (Compute_LWIR_Apparent_Radiance.py:9-20)
    f = lambda k: g[k].astype(np.float32)
    L32 = rt.compute_LWIR_apparent_radiance(f("X"), f("emis"), f("Ts"), f("tau"), f("La"), f("Ld"), dT=g["dT"])
    assert L32.dtype == np.float32 and rel_err(L32, g["L1"]) <= 5e-6 + TOL_L


# ------------------------------------------------------------------------------------------ G7 ILS
def test_g7_ils_golden(rt, golden):
    from radtxfr_amd import ILS_MAKO as ilsg
    g = golden("g7_ils.npz")
    X = np.linspace(float(g["X_lo"]), float(g["X_hi"]), int(g["X_n"]))
    Y2 = g["Y2"]
    xo, yo = rt.ILS_MAKO(X, Y2[:, 0])
    assert np.array_equal(xo, g["xo1"]) and yo.shape == g["yo1"].shape
    assert rel_err(yo, g["yo1"]) <= TOL_L
    xo, yo = rt.ILS_MAKO(X, Y2)
    assert yo.shape == g["yo2"].shape and rel_err(yo, g["yo2"]) <= TOL_L
    xo, yo = rt.ILS_MAKO(X, Y2, resFactor=2)
    assert np.array_equal(xo, g["xo3"]) and rel_err(yo, g["yo3"]) <= TOL_L
    yo = rt.ILS_MAKO(X, Y2, returnX=False, fwhm_sf=1.3, shift=0.4, scale=1.0005)
    assert rel_err(yo, g["yo4"]) <= TOL_L
    xg, yg = ilsg.ILS_MAKO(X, Y2[:, 0])
    assert np.array_equal(xg, g["xg1"])
    ok = np.isfinite(g["yg1"])
    assert np.array_equal(np.isfinite(yg), ok) and rel_err(yg[ok], g["yg1"][ok]) <= TOL_L
    xg, yg = ilsg.ILS_MAKO(X, Y2)
    ok = np.isfinite(g["yg2"])
    assert np.array_equal(np.isfinite(yg), ok) and rel_err(yg[ok], g["yg2"][ok]) <= TOL_L


def test_ils_many_spectra_and_edges(rt):
    rng = np.random.default_rng(11)
    X = np.linspace(760.0, 1320.0, 9000)
    Y = rng.uniform(0.0, 10.0, (X.size, 130)).astype(np.float32)  # column kernel, ragged nS
    xo, yo = rt.ILS_MAKO(X, Y)
    xr, yr = ref.ILS_MAKO(X, Y.astype(np.float64))
    assert np.array_equal(xo, xr) and yo.dtype == np.float32
    assert rel_err(yo, yr) <= TOL_L
    # coarse axis: some triangles catch no grid point -> NaN exactly where the reference has NaN (quirk 13)
    Xc = np.linspace(760.0, 1320.0, 40)
    Yc = rng.uniform(0, 1, Xc.size)
    xo, yo = rt.ILS_MAKO(Xc, Yc)
    with np.errstate(all="ignore"):
        xr, yr = ref.ILS_MAKO(Xc, Yc)
    assert np.array_equal(np.isnan(yo), np.isnan(yr))
    ok = ~np.isnan(yr)
    assert rel_err(yo[ok], yr[ok]) <= TOL_L


def test_ils_one_pass_over_large_spectra(rt):
    """The one-pass triangle ILS (ils_rows_kernel: every row of Y read once, partial sums per row chunk, fixed-order
    reduce) on a Y large enough to take it -- 300 000 rows x 256 spectra, 128 and 256 bands, default and widened
    triangles (more bands per chunk than one register pass holds; at fwhm_sf = 4 more than the workspace holds: the
    per-band fallback) -- against the reference's dense formulation on sampled columns."""
    import torch
    rng = np.random.default_rng(20261013)
    nx, nS = 300000, 256
    X = np.linspace(690.0, 1480.0, nx)
    Y = (rng.uniform(0.5, 10.0, (1, nS)) * (1.0 + 0.3 * np.sin(X[:, None] * rng.uniform(0.01, 0.5, (1, nS))))).astype(np.float32)
    Yd = torch.as_tensor(Y, device="cuda")
    cols = rng.choice(nS, 6, replace=False)
    for res, sf in ((None, 1.0), (2, 1.0), (None, 2.2), (None, 4.0)):
        xo, yo = rt.ILS_MAKO(X, Yd, resFactor=res, fwhm_sf=sf)
        yo = yo.cpu().numpy() if hasattr(yo, "cpu") else np.asarray(yo)
        xr, yr = ref.ILS_MAKO(X, Y[:, cols].astype(np.float64), resFactor=res, fwhm_sf=sf)
        assert np.array_equal(np.asarray(xo), xr)
        assert rel_err(yo[:, cols], yr) <= TOL_L, (res, sf)
    # the Gaussian variant (ILS_MAKO.py): ~28 bands reach every row; on a grid that covers all 128 band centres, and on
    # one that leaves some outside (their far tails average the grid's edge region)
    from radtxfr_amd import ILS_MAKO as ilsg
    for lo, hi in ((690.0, 1480.0), (800.0, 1250.0)):
        Xg = np.linspace(lo, hi, nx)
        xg, yg = ilsg.ILS_MAKO(Xg, Yd)
        yg = yg.cpu().numpy() if hasattr(yg, "cpu") else np.asarray(yg)
        with np.errstate(all="ignore"):
            xr, yr = ref.ILS_MAKO_gauss(Xg, Y[:, cols].astype(np.float64))
        assert np.array_equal(np.asarray(xg), xr)
        assert np.array_equal(np.isnan(yg[:, cols]), np.isnan(yr))
        ok = ~np.isnan(yr)
        assert rel_err(yg[:, cols][ok], yr[ok]) <= TOL_L, (lo, hi)


# ------------------------------------------------------------ C1 / C2 of BASELINE.json at full size
def test_c1_planck_beer_lambert(rt):
    """Config C1: X = linspace(700,1400,700), surface at 287.87 K with eps = 1, one layer at 287.87 K with
    OD = 0.3(1+sin(nu/20)); L = tau B(Ts) + (1-tau) B(T), through the shim's planckian + apparent radiance."""
    X = np.linspace(700.0, 1400.0, 700)
    OD = 0.3 * (1.0 + np.sin(X / 20.0))
    tau = np.exp(-OD)
    Bl = rt.planckian(X, 287.87)
    La = (1.0 - tau) * Bl
    L = rt.compute_LWIR_apparent_radiance(X, np.ones((X.size, 1)), np.array([287.87]), tau[:, None], La[:, None], np.zeros((X.size, 1)))
    want = tau * ref.planckian(X, 287.87) + (1.0 - tau) * ref.planckian(X, 287.87)
    assert L.shape == (700, 1, 1) and rel_err(L[:, 0, 0], want) <= TOL_L


def test_c2_voigt_full_config(hapi):
    """Config C2: 20 000-line H2O+CO2 table (seed 20261004), 700-1400 cm^-1 @ 0.01 (70 000 points),
    one layer at the surface state; the whole spectrum against the oracle."""
    tbl = synthetic.synth_line_table(synthetic.SEED_C2, 20000, 675.0, 1425.0)
    hapi.storage2cache_from_columns("c2", tbl)
    grid = np.linspace(700.0, 1400.0, 70000)
    T, p = 287.87, 100697.30225 / 101325.0
    om, xs = hapi.absorptionCoefficient_Voigt(SourceTables="c2", Environment={"T": T, "p": p}, OmegaGrid=grid)
    _, xr = ref.absorptionCoefficient_Voigt(tbl, T=T, p=p, OmegaGrid=grid)
    assert np.array_equal(om, grid) and rel_err(xs, xr) <= TOL_L
    # per-molecule components with HITRAN_units=False (what compute_OD sums), CO2 only
    _, xs2 = hapi.absorptionCoefficient_Voigt(Components=[(2, 1), (2, 2)], SourceTables="c2", Environment={"T": T, "p": p},
                                              OmegaGrid=grid, HITRAN_units=False)
    _, xr2 = ref.absorptionCoefficient_Voigt(tbl, Components=[(2, 1), (2, 2)], T=T, p=p, OmegaGrid=grid, HITRAN_units=False)
    assert rel_err(xs2, xr2) <= TOL_L


# ------------------------------------------------ C4: knots -> grid -> at-sensor radiance -> MAKO bands
def test_c4_band_radiance_vs_oracle(rt):
    """Config C4 at a size the oracle finishes in seconds: 40 emissivities on the ASTER-DB knots,
    760-1320 cm^-1 at 0.01 cm^-1, smooth synthetic TUD; device pipeline vs np.interp + the oracle's
    compute_LWIR_apparent_radiance + ILS_MAKO."""
    import torch
    from radtxfr_amd import engine, sensor
    Xe, em = synthetic.synth_emissivities(n_emis=40)
    grid = engine.Grid(760.0, 1320.0, 56000)
    X = grid.axis()
    tau = 0.5 + 0.45 * np.sin(X / 13.0)
    La = 2.0 + np.cos(X / 29.0)
    Ld = 4.0 + 2.0 * np.sin(X / 7.0)
    dev = torch.device("cuda")
    f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
    em_hi = sensor.interp_knots(grid, Xe, f32(em))
    em_ref = np.stack([np.interp(X, Xe, em[:, k]) for k in range(em.shape[1])], axis=1)
    assert rel_err(em_hi.cpu().numpy(), em_ref) <= 1e-6
    xo, Lb = sensor.band_radiance(grid, f32(tau), f32(La), f32(Ld), Xe, f32(em), 287.87)
    L_ref = ref.compute_LWIR_apparent_radiance(X, em_ref, np.array([287.87]), tau[:, None], La[:, None], Ld[:, None])[:, :, 0]
    xr, Lb_ref = ref.ILS_MAKO(X, L_ref)
    assert np.array_equal(xo, xr) and Lb.shape == Lb_ref.shape
    assert rel_err(Lb.cpu().numpy(), Lb_ref) <= TOL_L
    # fused form (no [nX][nE] array): same numbers, also at resFactor=2 and on an axis that sticks out of the knots
    xf, Lf = sensor.band_radiance_fused(grid, f32(tau), f32(La), f32(Ld), Xe, f32(em), 287.87)
    assert np.array_equal(xf, xr) and rel_err(Lf.cpu().numpy(), Lb_ref) <= TOL_L
    xr2, Lb_ref2 = ref.ILS_MAKO(X, L_ref, resFactor=2)
    xf2, Lf2 = sensor.band_radiance_fused(grid, f32(tau), f32(La), f32(Ld), Xe, f32(em), 287.87, resFactor=2)
    assert np.array_equal(xf2, xr2) and rel_err(Lf2.cpu().numpy(), Lb_ref2) <= TOL_L
    Xs = Xe[200:420]  # knots cover only 888..1108 cm^-1: np.interp holds the end values outside
    em_s = np.stack([np.interp(X, Xs, em[200:420, k]) for k in range(em.shape[1])], axis=1)
    L_s = ref.compute_LWIR_apparent_radiance(X, em_s, np.array([287.87]), tau[:, None], La[:, None], Ld[:, None])[:, :, 0]
    _, Lb_s = ref.ILS_MAKO(X, L_s)
    _, Lf_s = sensor.band_radiance_fused(grid, f32(tau), f32(La), f32(Ld), Xs, f32(em[200:420]), 287.87)
    assert rel_err(Lf_s.cpu().numpy(), Lb_s) <= TOL_L
    # np.interp end-value hold outside the knots
    g2 = engine.Grid(600.0, 700.0, 1001)
    e2 = sensor.interp_knots(g2, Xe, f32(em)).cpu().numpy()
    r2 = np.stack([np.interp(g2.axis(), Xe, em[:, k]) for k in range(em.shape[1])], axis=1)
    assert rel_err(e2, r2) <= 1e-6


# ----------------------------------------------------------------------------- C5: fused HSI cube
def test_c5_hsi_cube_vs_oracle(rt):
    """Config C5 at oracle size: 96 pixels, each a 2-of-6 endmember mixture with its own surface temperature
    (LWIR_HSI_Generator.py:151-167), monochromatic radiance -> rt.ILS_MAKO(resFactor=2). The device path never
    forms a per-pixel spectrum; the oracle does exactly that."""
    import torch
    from radtxfr_amd import engine, sensor
    Xe, em = synthetic.synth_emissivities(n_emis=2000)
    sc = synthetic.synth_scene(n_pix=96)
    E = em[:, sc["end_idx"]]  # [nk][6]
    grid = engine.Grid(760.0, 1320.0, 56000)
    X = grid.axis()
    tau = 0.5 + 0.45 * np.sin(X / 13.0)
    La = 2.0 + np.cos(X / 29.0)
    Ld = 4.0 + 2.0 * np.sin(X / 7.0)
    dev = torch.device("cuda")
    f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
    xo, cube = sensor.hsi_cube(grid, f32(tau), f32(La), f32(Ld), Xe, f32(E), torch.as_tensor(sc["kidx"], device=dev),
                               f32(sc["frac"]), torch.as_tensor(sc["T"], device=dev), resFactor=2)
    E_hi = np.stack([np.interp(X, Xe, E[:, k]) for k in range(E.shape[1])], axis=1)      # [nX][6]
    em_p = np.einsum("pm,xpm->xp", sc["frac"], E_hi[:, sc["kidx"]])                       # [nX][nPix]
    B = ref.planckian(X, sc["T"])                                                          # [nX][nPix]
    L = tau[:, None] * (em_p * B + (1 - em_p) * Ld[:, None]) + La[:, None]
    xr, Lr = ref.ILS_MAKO(X, L, resFactor=2)
    assert np.array_equal(xo, xr) and cube.shape == Lr.shape == (xr.size, 96)
    assert rel_err(cube.cpu().numpy(), Lr) <= TOL_L
    # a band-aligned shard computes the same numbers as the same rows of the full cube
    _, part = sensor.hsi_cube(grid, f32(tau), f32(La), f32(Ld), Xe, f32(E), torch.as_tensor(sc["kidx"], device=dev),
                              f32(sc["frac"]), torch.as_tensor(sc["T"], device=dev), resFactor=2, band_slice=(40, 97))
    assert torch.equal(part, cube[40:97])


# --------------------------------------------------------- size-independent properties at full size
def test_full_c3_width_properties():
    """5.5 M-point C3 grid x 32 layers at full size: properties that need no oracle run.
    (1) linearity: OD with all mixing ratios doubled = 2 x OD;  (2) a wavenumber shard computes the
    same numbers as the same slice of the full grid;  (3) tau = exp(-sum OD) consistency, 0<=tau<=1,
    0 <= L-up,L-down <= max Planck;  (4) a sub-window equals the oracle."""
    import torch
    from radtxfr_amd import engine
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    a = synthetic.c3_atmosphere(32)
    lines = engine.LineTable(full)
    grid = engine.Grid(500.0, 6000.0, 5500000)
    OD = engine.optical_depths(lines, grid, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    tau, Lu, Ld, _ = engine.tud(OD, grid, a["Ts"], a["Zs"])
    torch.cuda.synchronize()
    assert OD.shape == (32, 5500000) and bool(torch.isfinite(OD).all()) and float(OD.min()) >= 0.0
    # (3)
    s = OD.double().sum(0)
    assert float((tau[0].double() - torch.exp(-s)).abs().max()) <= TOL_TAU
    assert 0.0 <= float(tau.min()) and float(tau.max()) <= 1.0
    Bmax = float(ref.planckian(np.linspace(500.0, 6000.0, 5501), a["Ts"].max()).max())
    for v in (Lu[0], Ld):
        assert bool(torch.isfinite(v).all()) and float(v.min()) >= 0.0 and float(v.max()) <= Bmax * (1 + 1e-5)
    # (1) on a shard, which also checks (2)
    sh = grid.shard(2750000 - 1000, 40000 + 7)
    OD1 = engine.optical_depths(lines, sh, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    OD2 = engine.optical_depths(lines, sh, a["Ts"], a["Ps"], a["PLs"], 2.0 * a["MFs_VAL"], a["MFs_ID"])
    ref_slice = OD[:, sh.offset:sh.offset + sh.n]
    # an unaligned shard puts the rows elsewhere: other fp32 summation groups, and a point that was evaluated directly in
    # one tiling is interpolated from row nodes in the other (<= 2.3e-7 of a line's own contribution with a 2-row near zone)
    d = ((OD1 - ref_slice).abs() / ref_slice.abs().clamp_min(1e-3 * float(ref_slice.max()))).max()
    assert float(d) <= 3e-6, float(d)
    # a shard that starts and ends on tile boundaries reproduces the full grid's tiles, hence its bits
    from radtxfr_amd import _lib
    tp = int(_lib.load().rtx_voigt_tile_points())
    sh_al = grid.shard(2000 * tp, 31 * tp)
    OD_al = engine.optical_depths(lines, sh_al, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert torch.equal(OD_al, OD[:, sh_al.offset:sh_al.offset + sh_al.n]), "tile-aligned shard differs from the full grid"
    assert float(((OD2 - 2.0 * OD1).abs() / (2.0 * OD1).clamp_min(1e-30)).max()) <= 1e-6
    # run-to-run determinism: no atomics, fixed summation order -> the same bits every time
    OD1b = engine.optical_depths(lines, sh, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert torch.equal(OD1, OD1b), "the line-sum is not bit-reproducible from run to run"
    # (4) oracle on a 3000-point window of the full grid
    i0, n = 1234567, 3000
    Xw = grid.axis()[i0:i0 + n]
    sub = synthetic.subset_table(full, Xw[0] - 12.0, Xw[-1] + 12.0)
    ODr = np.stack([ref.layer_od(sub, Xw, a["Ts"][k], a["Ps"][k], a["PLs"][k], a["MFs_VAL"][k], a["MFs_ID"]) for k in range(32)], 1)
    tr, ur, dr = ref.tud_from_od(Xw, ODr, a["Ts"], a["Zs"])
    assert rel_err(OD[:, i0:i0 + n].T.double().cpu().numpy(), ODr) <= TOL_L
    assert np.max(np.abs(tau[0, i0:i0 + n].double().cpu().numpy() - tr)) <= TOL_TAU
    assert rel_err(Lu[0, i0:i0 + n].double().cpu().numpy(), ur) <= TOL_L
    assert rel_err(Ld[i0:i0 + n].double().cpu().numpy(), dr) <= TOL_L
    lines.close()


def test_full_c3_stratified_windows_vs_oracle():
    """The full C3 grid (5.5 M wavenumbers x 32 layers, the 100 000-line table) against the oracle on 14 windows of 1500
    points STRATIFIED over 500-6000 cm^-1 -- five of them inside the LWIR span 500-1500 cm^-1, the first and the last tile
    of the axis, a window across the middle tile boundary -- for the SURVEY-8d column (opaque almost everywhere) and the
    same column with mixing ratios x 1e-3 (tau spans (0, 1)): optical depth of all 32 layers, tau, L-up, L-down at the
    standing tolerances. (tests/acc_sweep.py as a driver-run test.)"""
    import torch
    from radtxfr_amd import engine
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    a = synthetic.c3_atmosphere(32)
    lines = engine.LineTable(full)
    grid = engine.Grid(500.0, 6000.0, 5500000)
    starts = [0, 123456, 400000, 700000, 998500, 1500000, 2100000, 2750000 - 750, 3300000, 3900000, 4500000, 5000000, 5400000,
              5500000 - 1500]
    n = 1500
    X = grid.axis()
    assert sum(1 for i0 in starts if X[i0 + n - 1] <= 1500.0) >= 4
    oracle = {}
    worst = {}
    for scale in (1.0, 1e-3):
        mf = a["MFs_VAL"] * scale
        OD = engine.optical_depths(lines, grid, a["Ts"], a["Ps"], a["PLs"], mf, a["MFs_ID"])
        tau, Lu, Ld, _ = engine.tud(OD, grid, a["Ts"], a["Zs"])
        torch.cuda.synchronize()
        e = [0.0, 0.0, 0.0, 0.0]
        spans = []
        for i0 in starts:
            Xw = X[i0:i0 + n]
            if i0 not in oracle:
                sub = synthetic.subset_table(full, Xw[0] - 12.0, Xw[-1] + 12.0)
                oracle[i0] = np.stack([ref.layer_od(sub, Xw, a["Ts"][k], a["Ps"][k], a["PLs"][k], a["MFs_VAL"][k], a["MFs_ID"])
                                       for k in range(32)], 1)
            ODr = oracle[i0] * scale  # optical depth is linear in the mixing ratios (same lines, same windows)
            tr, ur, dr = ref.tud_from_od(Xw, ODr, a["Ts"], a["Zs"])
            sl = slice(i0, i0 + n)
            e[0] = max(e[0], rel_err(OD[:, sl].T.double().cpu().numpy(), ODr))
            e[1] = max(e[1], float(np.max(np.abs(tau[0, sl].double().cpu().numpy() - tr))))
            e[2] = max(e[2], rel_err(Lu[0, sl].double().cpu().numpy(), ur))
            e[3] = max(e[3], rel_err(Ld[sl].double().cpu().numpy(), dr))
            spans.append(float(tr.max() - tr.min()))
        worst[scale] = e
        assert e[0] <= TOL_L and e[1] <= TOL_TAU and e[2] <= TOL_L and e[3] <= TOL_L, (scale, e)
        if scale < 1.0:
            assert max(spans) > 0.5  # the thinned column really exercises transmittances between 0 and 1
        del OD, tau, Lu, Ld
    print("stratified C3 sweep, worst [OD, |dtau|, Lu, Ld]:", worst)
    lines.close()


@pytest.mark.gpu
def test_smooth_and_reduce_resolution_vs_golden_g9(rt, golden):
    """rt.smooth / rt.reduceResolution (rtx_fir_reflect + rtx_cubic_resample, fp64) against the
    reference's own outputs; the spline is evaluated locally (cardinal spline), so agreement is ~1e-12, not bitwise."""
    g = golden("g9_reduce.npz")
    for args, key in (((11, "hanning"), "sm11"), ((50, "hamming"), "sm50"), ((7, "flat"), "smflat")):
        assert rel_err(rt.smooth(g["Y1"], *args), g[key]) < 1e-13
    Xo, Yo = rt.reduceResolution(g["Xf"], g["Y1"], 0.05)
    assert Xo.shape == g["Xo"].shape and np.allclose(Xo, g["Xo"], rtol=1e-13, atol=0)
    assert rel_err(Yo, g["Yo1"]) < 1e-10
    Y2o = rt.reduceResolution(g["Xf"], g["Y2"], 0.05, X_out=g["Xo"])
    assert Y2o.shape == g["Yo2"].shape and rel_err(Y2o, g["Yo2"]) < 1e-10
    Xo8, Yo8 = rt.reduceResolution(g["Xf"], g["Y1"], 0.02, N=8, window="blackman")
    assert Xo8.shape == g["Xo8"].shape and rel_err(Yo8, g["Yo8"]) < 1e-10
    # the reference's early returns
    assert rt.smooth(g["Y1"][:5], 11) is not None and rt.smooth(g["Y1"][:5], 11).shape == (5,)
    with pytest.raises(NotImplementedError):
        rt.reduceResolution(g["Xf"], g["Y1"], 0.05, X_out=g["Xf"][:10])  # inside the distorted end region


@pytest.mark.gpu
def test_reduce_resolution_full_size_float32_device_path():
    """The caller's configuration (Generate_LWIR_TUD.py:76-85): 690-1410 cm^-1 at 0.0005 -> 0.25 cm^-1, N = 4, on
    device-resident float32 rows as rtx_tud writes them; checked against the oracle on a window and through
    properties at full size (constants are preserved, output axis, linearity)."""
    import torch
    from radtxfr_amd import engine

    n = 1440001
    x0, h = 690.0, 720.0 / (n - 1)
    X = x0 + h * np.arange(n)
    rng = np.random.default_rng(11)
    Y = (1.0 + 0.3 * np.sin(3.0 * X) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    rows = torch.as_tensor(np.stack([Y, np.ones_like(Y), 2.0 * Y + 1.0]), device="cuda")
    xo, out = engine.reduce_resolution(rows, x0, h, n, 0.25)
    out = out.cpu().numpy()
    assert xo.size == 11513 and abs(xo[0] - X[500]) < 1e-9 and abs(xo[-1] - X[-501]) < 1e-9
    assert np.max(np.abs(out[1] - 1.0)) < 1e-12                      # a constant stays constant
    assert np.max(np.abs(out[2] - (2.0 * out[0] + 1.0))) < 1e-6      # linear in Y (inputs are float32)
    # oracle on a window well inside (the smoother reaches 500 samples, the spline ~40)
    lo, hi = 400000, 440000
    xw, yw = ref.reduceResolution(X[lo:hi], Y[lo:hi].astype(np.float64), 0.25)
    sel = (xo >= xw[0]) & (xo <= xw[-1])
    want = ref.reduceResolution(X[lo:hi], Y[lo:hi].astype(np.float64), 0.25, X_out=xo[sel])
    assert rel_err(out[0][sel], want) < 1e-10


# --------------------------------------------------------------------- G10 Lorentz / Doppler
def test_g10_lorentz_doppler_golden(hapi, golden):
    """hapi.absorptionCoefficient_Lorentz / _Doppler (rtx_line_prep_profile + the Voigt line-sum kernels) against the
    reference run. Lorentz is exactly the line-sum's far-wing rational; Doppler is the Voigt profile at y = 0."""
    g = golden("g10_lorentz_doppler.npz")
    tbl = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    hapi.storage2cache_from_columns("g10", tbl)
    gl = np.linspace(float(g["gl_lo"]), float(g["gl_hi"]), int(g["gl_n"]))
    gd = np.linspace(float(g["gd_lo"]), float(g["gd_hi"]), int(g["gd_n"]))
    for tag in ("l01", "l32"):
        om, xs = hapi.absorptionCoefficient_Lorentz(SourceTables="g10", Environment={"T": float(g["T_" + tag]), "p": float(g["p_" + tag])},
                                                    OmegaGrid=gl)
        assert np.array_equal(om, gl) and rel_err(xs, g["lor_" + tag]) <= TOL_L, tag
    _, xs = hapi.absorptionCoefficient_Lorentz(Components=[(1, 1), (2, 1, 0.5)], SourceTables="g10", Environment={"T": 250.0, "p": 0.4},
                                               OmegaGrid=gl[5000:12000], HITRAN_units=False, OmegaWing=1.0, OmegaWingHW=20.0,
                                               Diluent={"air": 0.7, "self": 0.3})
    assert rel_err(xs, g["lor_opt"]) <= TOL_L
    for tag, (Tk, pk) in (("a", (296.0, 1.0)), ("b", (220.0, 0.05))):
        _, xs = hapi.absorptionCoefficient_Doppler(SourceTables="g10", Environment={"T": Tk, "p": pk}, OmegaGrid=gd)
        assert rel_err(xs, g["dop_" + tag]) <= TOL_L, tag
    _, xs = hapi.absorptionCoefficient_Gauss(SourceTables="g10", Environment={"T": 296.0, "p": 1.0}, OmegaGrid=gd, LineShift=False,
                                             HITRAN_units=False, OmegaWing=0.05)
    assert rel_err(xs, g["dop_noshift"]) <= TOL_L


def test_sdvoigt_is_the_voigt_limit_without_sd_columns(hapi, golden):
    """hapi.absorptionCoefficient_SDVoigt on a table without SD_* columns is the Voigt profile (misc/hapi.py:9908-9915):
    same result as the Voigt shim, within tolerance of the reference's Voigt golden."""
    g = golden("g4_voigt_xsec.npz")
    grid = _g4_table(hapi, g)[20000:26000]
    env = {"T": float(g["T_l01"]), "p": float(g["p_l01"])}
    _, xv = hapi.absorptionCoefficient_Voigt(SourceTables="g4", Environment=env, OmegaGrid=grid)
    _, xs = hapi.absorptionCoefficient_SDVoigt(SourceTables="g4", Environment=env, OmegaGrid=grid)
    assert np.array_equal(xs, xv)
    assert rel_err(xs, g["xs_l01"][20000:26000]) <= TOL_L


def test_g11_sdvoigt_golden(hapi, golden):
    """hapi.absorptionCoefficient_SDVoigt with speed-dependence columns (rtx_sdvoigt_sum: pcqsdhc PART1-4 in fp64) against
    the reference run; fp64 throughout, so agreement is at rounding level, not the fp32 tolerance."""
    g = golden("g11_sdvoigt.npz")
    tbl = dict(synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"])))
    tbl["SD_air"], tbl["SD_self"] = g["SD_air"], g["SD_self"]
    hapi.storage2cache_from_columns("g11", tbl)
    grid = np.linspace(float(g["g_lo"]), float(g["g_hi"]), int(g["g_n"]))
    cases = (("a", dict(Environment={"T": 250.0, "p": 0.3})),
             ("b", dict(Environment={"T": 296.0, "p": 1.0}, Diluent={"air": 0.6, "self": 0.4})),
             ("c", dict(Environment={"T": 220.0, "p": 0.01})),
             ("d", dict(Environment={"T": 300.0, "p": 0.8}, HITRAN_units=False, OmegaWing=0.5, OmegaWingHW=20.0,
                        Components=[(1, 1), (2, 1, 0.5)])))
    for tag, kw in cases:
        om, xs = hapi.absorptionCoefficient_SDVoigt(SourceTables="g11", OmegaGrid=grid, **kw)
        assert np.array_equal(om, grid) and rel_err(xs, g["xs_" + tag]) <= 1e-9, tag
    # the reference's default alias absorptionCoefficient = absorptionCoefficient_HT is the same thing for tables
    # without Hartmann-Tran columns (misc/hapi.py:10505-10640, 11377); HT columns are refused
    _, xh = hapi.absorptionCoefficient(SourceTables="g11", OmegaGrid=grid, Environment={"T": 250.0, "p": 0.3})
    assert rel_err(xh, g["xs_a"]) <= 1e-9
    ht = dict(tbl)
    ht["gamma_HT_0_air_296"] = np.full(int(g["n_lines"]), 0.05)
    hapi.storage2cache_from_columns("g11h", ht)
    with pytest.raises(NotImplementedError):
        hapi.absorptionCoefficient_HT(SourceTables="g11h", OmegaGrid=grid)
    # a tiny speed dependence (PART2: |X| <= 3e-8 |Y|) tends to the Voigt profile
    tiny = dict(tbl)
    tiny["SD_air"] = np.full(int(g["n_lines"]), 1e-9)
    tiny.pop("SD_self")
    hapi.storage2cache_from_columns("g11t", tiny)
    _, xt = hapi.absorptionCoefficient_SDVoigt(SourceTables="g11t", OmegaGrid=grid, Environment={"T": 250.0, "p": 0.3})
    _, xo = ref.absorptionCoefficient_SDVoigt(tiny, T=250.0, p=0.3, OmegaGrid=grid)
    _, xv = ref.absorptionCoefficient_Voigt(tiny, T=250.0, p=0.3, OmegaGrid=grid)
    assert rel_err(xt, xo) <= 1e-7 and rel_err(xt, xv) <= 1e-6


def test_cross_sections_at_the_reference_callers_settings(hapi):
    """The reference's one in-tree hapi caller (misc/RT_gen_AbsXS_files.py:15-18, 86-92) runs absorptionCoefficient_SDVoigt
    with WavenumberStep = 0.0025 and WavenumberWingHW = 350: windows of +-350 gamma0 = +-25 ... 38 cm^-1 at one atmosphere,
    seven times wider than the default 50. A 200 cm^-1 slice of that configuration (80 001 points, 2500 lines with wings
    reaching in from 40 cm^-1 outside) against the oracle, for a table without speed-dependence columns (Gamma2 = 0: pcqsdhc
    PART1, the fp32 Voigt line-sum with its row-level nodes on windows this wide) and with SD_air (PART4 in fp64, almost
    all of it the closed-form far-wing difference), at the surface state and at 0.05 atm."""
    from radtxfr_amd import afit_xs
    tbl = dict(synthetic.synth_line_table(77, 2500, 760.0, 1040.0))
    X = np.linspace(800.0, 1000.0, 80001)
    hapi.storage2cache_from_columns("xs350", tbl)
    for T, p in ((296.0, 1.0), (260.0, 0.05)):
        _, xv = hapi.absorptionCoefficient_SDVoigt(SourceTables="xs350", Environment={"T": T, "p": p}, WavenumberGrid=None, OmegaGrid=X,
                                                   WavenumberWingHW=350.0)
        _, xr = ref.absorptionCoefficient_Voigt(tbl, T=T, p=p, OmegaGrid=X, OmegaWingHW=350.0)
        assert rel_err(xv, xr) <= TOL_L, (T, p, rel_err(xv, xr))
    sd = dict(tbl)
    sd["SD_air"] = np.round(np.random.default_rng(78).uniform(0.05, 0.2, 2500), 3)
    hapi.storage2cache_from_columns("xs350sd", sd)
    for T, p in ((296.0, 1.0), (260.0, 0.05)):
        _, xs = hapi.absorptionCoefficient_SDVoigt(SourceTables="xs350sd", Environment={"T": T, "p": p}, OmegaGrid=X, WavenumberWingHW=350.0)
        _, xr = ref.absorptionCoefficient_SDVoigt(sd, T=T, p=p, OmegaGrid=X, OmegaWingHW=350.0)
        assert rel_err(xs, xr) <= 1e-9, (T, p, rel_err(xs, xr))
    # the batched T x p generator on the same table equals the per-state calls
    xs_grid = afit_xs.cross_section_grid("xs350sd", [296.0], [1.0], X, WavenumberWingHW=350.0)
    _, one = hapi.absorptionCoefficient_SDVoigt(SourceTables="xs350sd", Environment={"T": 296.0, "p": 1.0}, OmegaGrid=X, WavenumberWingHW=350.0)
    assert rel_err(xs_grid[0, 0], one) <= 1e-12
    for n in ("xs350", "xs350sd"):
        hapi.LOCAL_TABLE_CACHE.pop(n)


def test_sdvoigt_node_levels_equal_point_by_point(hapi):
    """rtx_sdvoigt_sum's default kernel (far wings at 32 tile nodes / 12 row nodes in fp64, csrc/rtx_sdvoigt.hip) against its
    point-by-point cross-check (RADTXFR_SD_KERNEL=gather) on a table that mixes every regime of pcqsdhc: SD = 0 (PART1), a
    tiny speed dependence (PART2 over the whole window), ordinary values (PART4, closed-form far wing) and a large one
    (Gamma0 close to 1.5 Gamma2), at the surface, at 0.01 atm and at 1e-4 atm, with wings of 350 and of 50
    half-widths, on a grid whose last tile is ragged. The node levels may differ from the point-by-point sum by their
    interpolation bound (1e-10 of each line's own positive contribution)."""
    n = 3000
    tbl = dict(synthetic.synth_line_table(91, n, 760.0, 1040.0))
    rng = np.random.default_rng(92)
    sd = np.round(rng.uniform(0.05, 0.2, n), 3)
    kind = rng.integers(0, 8, n)
    sd[kind == 0] = 0.0
    sd[kind == 1] = 1e-9
    sd[kind == 2] = 0.6
    sd[kind == 3] = 1e-5
    tbl["SD_air"] = sd
    hapi.storage2cache_from_columns("sdmix", tbl)
    X = np.linspace(800.0, 1000.0, 80001)
    worst = 0.0
    for T, p, hw in ((296.0, 1.0, 350.0), (230.0, 0.01, 350.0), (250.0, 1e-4, 350.0), (296.0, 0.5, 50.0)):
        kw = dict(SourceTables="sdmix", Environment={"T": T, "p": p}, OmegaGrid=X, WavenumberWingHW=hw)
        _, xs = hapi.absorptionCoefficient_SDVoigt(**kw)
        os.environ["RADTXFR_SD_KERNEL"] = "gather"
        try:
            _, xg = hapi.absorptionCoefficient_SDVoigt(**kw)
        finally:
            del os.environ["RADTXFR_SD_KERNEL"]
        nz = xg != 0.0
        assert nz.any() and np.all(xg >= 0.0) and np.array_equal(xs[~nz], xg[~nz])
        e = float(np.max(np.abs(xs[nz] - xg[nz]) / xg[nz]))
        worst = max(worst, e)
        assert e <= 1e-9, (T, p, hw, e)
    # ragged and tiny grids, and a wavenumber shard with an offset (engine API): one point, one row + 1, one tile + 1
    import torch
    from radtxfr_amd import engine
    lines = hapi._device_table(["sdmix"])
    w = np.ones((len(lines.species), 1))
    for n_tot, off, cnt in ((2, 0, 2), (2, 1, 1), (65, 0, 65), (40001, 0, 1025), (40001, 30000, 5000), (40001, 39999, 2)):
        grid = engine.Grid(900.0, 1000.0, n_tot).shard(off, cnt)
        outs = []
        for kern in ("", "gather"):
            if kern:
                os.environ["RADTXFR_SD_KERNEL"] = kern
            try:
                o = torch.empty((1, cnt), dtype=torch.float64, device="cuda")
                engine.voigt_sum(lines, grid, [296.0], [1.0], w, out_f64=o, omega_wing_hw=350.0, scale=2.0 ** 70, profile=3)
                outs.append(o.cpu().numpy()[0])
            finally:
                os.environ.pop("RADTXFR_SD_KERNEL", None)
        assert np.all(outs[1] > 0.0) and float(np.max(np.abs(outs[0] - outs[1]) / outs[1])) <= 1e-9, (n_tot, off, cnt)
    hapi.LOCAL_TABLE_CACHE.pop("sdmix")


def test_afit_xs_grid_batched_states(hapi, tmp_path):
    """afit_xs.cross_section_grid / generate_xs_files (the T x p loop of misc/RT_gen_AbsXS_files.py:86-92 as one
    batched launch): every state equals the per-state hapi shim call and agrees with the oracle; files round-trip."""
    from radtxfr_amd import afit_xs

    tbl = synthetic.synth_line_table(11, 400, 880.0, 930.0)
    h2o = {k: np.asarray(v)[np.asarray(tbl["molec_id"]) == 1] for k, v in tbl.items()}
    hapi.storage2cache_from_columns("H2O", h2o)
    X = np.linspace(900.0, 910.0, 4001)  # 0.0025 cm^-1, the generator's step
    T = np.array([275.0, 300.0, 320.0])
    P = np.array([0.85, 1.05])
    xs = afit_xs.cross_section_grid("H2O", T, P, X, WavenumberWingHW=350.0)
    assert xs.shape == (3, 2, 4001)
    for it, t in enumerate(T):
        for ip, p in enumerate(P):
            _, one = hapi.absorptionCoefficient_SDVoigt(SourceTables="H2O", HITRAN_units=True, Environment={"T": t, "p": p},
                                                        OmegaGrid=X, IntensityThreshold=0, WavenumberWingHW=350)
            assert rel_err(xs[it, ip], one) <= 1e-6
    _, want = ref.absorptionCoefficient_Voigt(h2o, T=300.0, p=1.05, OmegaGrid=X, OmegaWingHW=350.0)
    assert rel_err(xs[1, 1], want) <= TOL_L
    names = afit_xs.generate_xs_files("H2O", 1, T, P, X, "synthetic", WavenumberWingHW=350.0, directory=str(tmp_path))
    assert len(names) == 6 and os.path.basename(names[0]) == "XS-01-0275K-086126Pa.bin"
    back = afit_xs.AFIT_XS_read(names[3])  # T = 300, p = 1.05
    assert back["T"] == 300.0 and abs(back["P"] - 101325 * 1.05) < 1e-6 and np.array_equal(back["Y"], xs[1, 1])


# --------------------------------------------------------------------- alternative line-sum formulations
@pytest.mark.parametrize("kernel", ["scatter"])
def test_alternative_line_sum_kernels_agree(kernel):
    """RADTXFR_VOIGT_KERNEL=scatter (every row point by point: the cross-check of the default kernel's node interpolation)
    gives the same layer optical depths: a child process (the choice is read once per process), compared with the default
    formulation computed here and with the oracle."""
    import subprocess
    import sys
    import tempfile

    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from radtxfr_amd import synthetic, radiative_transfer as rt\n"
        "full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)\n"
        "sub = synthetic.subset_table(full, 988.0, 1016.0)\n"
        "a = synthetic.c3_atmosphere(32)\n"
        "od = np.stack([rt.compute_OD(1000.0, 1004.0, DVOUT=0.001, line_table=sub, T=a['Ts'][k], P=a['Ps'][k], PL=a['PLs'][k],\n"
        "                             MF_VAL=a['MFs_VAL'][k], MF_ID=a['MFs_ID'])[1] for k in (0, 31)], 1)\n"
        "np.save(sys.argv[1], od)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "od.npy")
        env = dict(os.environ, RADTXFR_VOIGT_KERNEL=kernel)
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
        od_alt = np.load(out)
    from radtxfr_amd import radiative_transfer as rt

    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    sub = synthetic.subset_table(full, 988.0, 1016.0)
    a = synthetic.c3_atmosphere(32)
    X = None
    cols = []
    want = []
    for k in (0, 31):
        X, od = rt.compute_OD(1000.0, 1004.0, DVOUT=0.001, line_table=sub, T=a["Ts"][k], P=a["Ps"][k], PL=a["PLs"][k],
                              MF_VAL=a["MFs_VAL"][k], MF_ID=a["MFs_ID"])
        cols.append(od)
        want.append(ref.layer_od(sub, X, a["Ts"][k], a["Ps"][k], a["PLs"][k], a["MFs_VAL"][k], a["MFs_ID"]))
    od_def = np.stack(cols, 1)
    assert rel_err(od_alt, od_def) <= 2e-6, kernel     # formulations differ only in fp32 summation order / interpolation
    assert rel_err(od_alt, np.stack(want, 1)) <= TOL_L, kernel


# --------------------------------------------------------------------- TUD: randomised configurations
def test_tud_random_configurations_vs_oracle():
    """rtx_tud against the oracle on random columns: 1..70 layers, several sensor altitudes (also non-monotone ones and
    ones below the surface layer; ascending height grids -- one recurrence per slant, snapshots per altitude -- and
    shuffled ones -- pairs in blocks), slant paths, 1..40 angles (more than one block of streams), returnOD, and optical
    depths whose magnitude changes by layer and along the spectrum so that waves are thin, thick, mixed and opaque --
    the paths the kernel selects per (wave, layer), the opaque-slab start and the staged OD loads."""
    import torch
    from radtxfr_amd import engine

    rng = np.random.default_rng(20261011)
    for trial in range(24):
        nL = int(rng.integers(1, 71))
        n = int(rng.integers(700, 3000))
        Z = np.sort(rng.uniform(0.0, 60.0, nL))
        if trial % 4 == 2:
            Z = rng.permutation(Z)  # a height grid that is not ascending: the Z <= zs masks are no prefixes (tau: mask, L-up: count)
        T = rng.uniform(190.0, 310.0, nL)
        lo = float(rng.uniform(500.0, 5500.0))
        grid = engine.Grid(lo, lo + 2.0, n)
        X = grid.axis()
        # per-layer scale 10^U(-6, 2), smooth spectral structure x a few narrow spikes, some exactly zero columns
        scale = 10.0 ** rng.uniform(-6.0, 2.0, nL)
        shape = 1.0 + 0.9 * np.sin(rng.uniform(5, 60) * X + rng.uniform(0, 6))
        for _ in range(4):
            c, w = rng.uniform(X[0], X[-1]), 10.0 ** rng.uniform(-3.0, -1.0)
            shape = shape + rng.uniform(10, 3000) * w * w / ((X - c) ** 2 + w * w)
        OD = (scale[:, None] * shape[None, :]).astype(np.float32)
        if trial % 3 == 0:
            OD[:, rng.integers(0, n, 20)] = 0.0
        nalt = int(rng.integers(1, 6))
        alts = rng.uniform(-1.0, 70.0, nalt)
        if trial % 4 == 1:
            alts = np.array([500.0])
        theta = float(rng.choice([0.0, 0.3, 1.1]))
        if trial % 3 == 1:
            theta = np.array([0.0, 0.5, 1.2])  # with up to 5 altitudes: more (altitude, slant) pairs than one block
        nA = int(rng.choice([1, 2, 7, 30, 33, 40]))
        ret_od = bool(trial % 5 == 2)
        tau, Lu, Ld, (nZ, nMu) = engine.tud(torch.as_tensor(OD, device="cuda"), grid, T, Z, Altitudes=alts, theta_r=theta,
                                            N_angle=nA, returnOD=ret_od)
        tr, ur, dr = ref.tud_from_od(X, OD.astype(np.float64).T, T, Z, Altitudes=alts, theta_r=theta, N_angle=nA, returnOD=ret_od)
        tau_h = tau.double().cpu().numpy().reshape(nZ, nMu, -1).transpose(2, 0, 1).reshape(np.shape(tr))
        Lu_h = Lu.double().cpu().numpy().reshape(nZ, nMu, -1).transpose(2, 0, 1).reshape(np.shape(ur))
        tag = (trial, nL, nalt, str(theta), nA, ret_od)
        if ret_od:
            assert rel_err(tau_h, tr) <= TOL_L, tag
        else:
            assert np.max(np.abs(tau_h - tr)) <= TOL_TAU, tag
        assert rel_err(Lu_h, ur) <= TOL_L, tag
        if nA == 1:  # the reference divides 0 by 0 (theta = 0 has weight 0): NaN here too
            assert np.isnan(dr).all() and bool(torch.isnan(Ld).all()), tag
        else:
            assert rel_err(Ld.double().cpu().numpy(), dr) <= TOL_L, tag
        # the downwelling above came from the angle-summed form (tud_g_kernel); asking for the per-stream radiances runs
        # the stream kernel (column resident in LDS up to 36 layers, chunked beyond): same tau / L-up / L-down, and the
        # streams themselves against the reference's own recurrence per angle
        t2, u2, Ld2, _, Ld_ang = engine.tud(torch.as_tensor(OD, device="cuda"), grid, T, Z, Altitudes=alts, theta_r=theta,
                                            N_angle=nA, returnOD=ret_od, per_angle=True)
        if ret_od:
            assert rel_err(t2.double().cpu().numpy(), tau.double().cpu().numpy()) <= 2e-6, tag
        else:  # (the two kernels add the column's OD in opposite orders)
            assert float((t2 - tau).abs().max()) <= TOL_TAU, tag
        assert rel_err(u2.double().cpu().numpy(), Lu.double().cpu().numpy()) <= 2e-6, tag
        if nA > 1:
            assert rel_err(Ld2.double().cpu().numpy(), dr) <= TOL_L, tag
            assert rel_err(Ld2.double().cpu().numpy(), Ld.double().cpu().numpy()) <= 3e-6, tag
        nd = int((Z <= np.asarray(alts).ravel()[-1]).sum())
        Bk = ref.planckian(X, T)                      # (n, nL)
        for q in rng.choice(nA, size=min(nA, 3), replace=False):
            sec = 1.0 / np.cos(q * (np.pi / 2) / nA)
            Lq = np.zeros(n)
            for j in range(nd - 1, -1, -1):
                t = np.exp(-OD[j].astype(np.float64) * sec)
                Lq = t * Lq + (1.0 - t) * Bk[:, j]
            assert rel_err(Ld_ang[q].double().cpu().numpy(), Lq) <= TOL_L, (tag, int(q))


def test_tud_downwelling_table_extremes():
    """The angle-summed downwelling (tud_g_kernel) at the ends of its table: columns from 1e-9 to 1e4 total optical
    depth (thin layers over thick ones and the reverse), an all-zero column (exactly 0), a NaN depth (NaN like the
    reference's recurrence), slightly negative depths (rounding noise of a caller's own OD), N_angle from 2 to 96."""
    import torch
    from radtxfr_amd import engine

    rng = np.random.default_rng(20261012)
    nL, n = 32, 1024
    Z = np.linspace(0.0, 9.5, nL)
    T = np.linspace(288.0, 226.0, nL)
    grid = engine.Grid(800.0, 801.0, n)
    X = grid.axis()
    for nA in (2, 3, 30, 96):
        OD = np.empty((nL, n), dtype=np.float32)
        tot = 10.0 ** np.linspace(-9.0, 4.0, n)                     # total depth of column i
        prof = rng.dirichlet(np.full(nL, 0.3), size=n).T            # how it is spread over the layers: very uneven
        OD[:] = (tot[None, :] * prof).astype(np.float32)
        OD[:, 5] = 0.0
        OD[:, 7] = -1e-7 * rng.uniform(0, 1, nL)
        tau, Lu, Ld, _ = engine.tud(torch.as_tensor(OD, device="cuda"), grid, T, Z, N_angle=nA)
        tr, ur, dr = ref.tud_from_od(X, OD.astype(np.float64).T, T, Z, N_angle=nA)
        Ld_h = Ld.double().cpu().numpy()
        assert Ld_h[5] == 0.0
        # relative to each column's own radiance (no floor: the thin columns carry 1e-9 of the thick ones' radiance)
        ok = np.abs(dr) > 0
        assert np.max(np.abs(Ld_h[ok] - dr[ok]) / np.abs(dr[ok])) <= TOL_L, nA
        assert rel_err(Lu.double().cpu().numpy().ravel(), ur) <= TOL_L
        OD[3, 11] = np.nan
        tau, Lu, Ld, _ = engine.tud(torch.as_tensor(OD, device="cuda"), grid, T, Z, N_angle=nA)
        assert bool(torch.isnan(Ld[11])) and not bool(torch.isnan(Ld[12]))


# --------------------------------------------------------------------- line-sum: randomised grids and states
def test_line_sum_random_grids_vs_oracle(hapi):
    """hapi.absorptionCoefficient_Voigt / _Lorentz against the oracle for random grid steps (3e-4 .. 2e-2 cm^-1: Lorentz
    widths from hundreds of grid points down to a few, i.e. every mix of tile-level, row-level, near and band rows),
    window placements (ragged last tiles, grids shorter than a tile), temperatures, pressures from 1e-3 to 1.2 atm
    (y from << 1, the fp64 pass, to >> 15, no band at all), wing settings and diluent mixes."""
    rng = np.random.default_rng(20261012)
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    for trial in range(10):
        step = float(10.0 ** rng.uniform(-3.5, -1.7))
        n = int(rng.integers(300, 9000))
        lo = float(rng.uniform(500.0, 5990.0 - n * step))
        grid = np.linspace(lo, lo + (n - 1) * step, n)
        p = float(10.0 ** rng.uniform(-3.0, 0.08))
        Tk = float(rng.uniform(190.0, 320.0))
        hw = float(rng.choice([5.0, 25.0, 50.0]))
        wing = float(rng.choice([0.0, 0.0, 0.7]))
        reach = max(wing, hw * 0.12 * max(p, 0.02)) + 1.0
        sub = synthetic.subset_table(full, grid[0] - reach, grid[-1] + reach)
        name = "fz%d" % trial
        hapi.storage2cache_from_columns(name, sub)
        dil = {"air": 0.8, "self": 0.2} if trial % 3 == 0 else {}
        kw = dict(SourceTables=name, Environment={"T": Tk, "p": p}, OmegaGrid=grid, OmegaWing=wing, OmegaWingHW=hw, Diluent=dil)
        okw = dict(T=Tk, p=p, OmegaGrid=grid, OmegaWing=wing, OmegaWingHW=hw, Diluent=dil or None)
        tag = (trial, step, n, lo, p, Tk, hw, wing)
        _, xs = hapi.absorptionCoefficient_Voigt(**kw)
        _, want = ref.absorptionCoefficient_Voigt(sub, **okw)
        assert rel_err(xs, want) <= TOL_L, ("voigt",) + tag
        if trial % 2 == 0:
            _, xs = hapi.absorptionCoefficient_Lorentz(**kw)
            _, want = ref.absorptionCoefficient_Lorentz(sub, **okw)
            assert rel_err(xs, want) <= TOL_L, ("lorentz",) + tag


# --------------------------------------------------------------------- streaming stages: randomised shapes
def test_streaming_stages_random_shapes_vs_oracle(rt):
    """planckian / brightnessTemperature / BT2L, compute_LWIR_apparent_radiance and ILS_MAKO (both variants) against
    the oracle for random shapes: odd sample counts (the float4 kernels' remainders), 1..9 atmospheres and surface
    temperatures, with and without dT / return_Ls, grids that clip the band list at either end."""
    from radtxfr_amd import ILS_MAKO as ilsg

    rng = np.random.default_rng(20261013)
    for trial in range(8):
        nX = int(rng.integers(40, 400))
        X = np.sort(rng.uniform(700.0, 1400.0, nX))
        nE, nA, nT = int(rng.integers(1, 12)), int(rng.integers(1, 10)), int(rng.integers(1, 6))
        emis = rng.uniform(0.0, 1.0, (nX, nE))
        Ts = rng.uniform(250.0, 330.0, nA)
        tau = rng.uniform(0.0, 1.0, (nX, nA))
        La = rng.uniform(0.0, 5.0, (nX, nA))
        Ld = rng.uniform(0.0, 9.0, (nX, nA))
        dT = rng.uniform(-10.0, 10.0, nT) if trial % 2 else None
        want = ref.compute_LWIR_apparent_radiance(X, emis, Ts, tau, La, Ld, dT=dT, return_Ls=True)
        got = rt.compute_LWIR_apparent_radiance(X, emis, Ts, tau, La, Ld, dT=dT, return_Ls=True)
        for g_, w_ in zip(got, want):
            assert g_.shape == w_.shape and rel_err(g_, w_) <= TOL_L, (trial, nX, nE, nA, nT)
        # Planck family (fp64 entry points)
        Tm = rng.uniform(180.0, 340.0, (int(rng.integers(1, 5)), int(rng.integers(1, 4))))
        L = rt.planckian(X, Tm)
        assert rel_err(L, ref.planckian(X, Tm)) <= 1e-12
        assert rel_err(rt.brightnessTemperature(X, L), ref.brightnessTemperature(X, L)) <= 1e-10
        assert rel_err(rt.BT2L(X, np.broadcast_to(Tm[None], (nX,) + Tm.shape)), ref.BT2L(X, np.broadcast_to(Tm[None], (nX,) + Tm.shape))) <= 1e-12
    for trial in range(6):
        n = int(rng.integers(3000, 30000))
        lo = float(rng.uniform(700.0, 800.0))
        hi = float(rng.uniform(1150.0, 1400.0))
        X = np.linspace(lo, hi, n)
        nS = int(rng.choice([1, 2, 3, 5, 16, 17, 33, 70]))
        Y = rng.uniform(0.0, 10.0, (n, nS)) * (1.0 + 0.5 * np.sin(0.05 * X))[:, None]
        Yin = Y[:, 0] if nS == 1 and trial % 2 else Y
        xo, yo = rt.ILS_MAKO(X, Yin, fwhm_sf=float(rng.uniform(0.7, 1.6)) if trial % 3 else 1.0)
        xr, yr = ref.ILS_MAKO(X, Yin, fwhm_sf=1.0) if not trial % 3 else (None, None)
        if xr is not None:
            assert np.array_equal(xo, xr) and yo.shape == yr.shape and rel_err(yo, yr) <= TOL_L, (trial, n, nS)
        xg, yg = ilsg.ILS_MAKO(X, Yin)
        xgr, ygr = ref.ILS_MAKO_gauss(X, Yin)
        assert np.array_equal(xg, xgr)
        # the Gaussian band list is not clipped to the grid (ILS_MAKO.py:13-19): a band centred outside is the average
        # of the edge region under the far tail until the reference's fp64 weights all underflow (0/0 = NaN). Compare
        # everything except the bands whose largest weight is a denormal (exponent between -700 and -745).
        sig = np.abs(np.gradient(xgr))
        d0 = np.maximum(np.maximum(X[0] - xgr, xgr - X[-1]), 0.0)
        expo = 0.5 * (d0 / sig) ** 2 + np.log(sig * np.sqrt(2.0 * np.pi))
        live, dead = expo < 700.0, expo > 745.5
        yg2, ygr2 = yg.reshape(xgr.size, -1), ygr.reshape(xgr.size, -1)
        assert np.isfinite(ygr2[live]).all() and np.isfinite(yg2[live]).all(), (trial, n, nS)
        assert rel_err(yg2[live], ygr2[live]) <= TOL_L, (trial, n, nS)
        assert np.isnan(ygr2[dead]).all() and np.isnan(yg2[dead]).all(), (trial, n, nS)


def test_fused_sensor_paths_random_configurations_vs_oracle(rt):
    """sensor.band_radiance / band_radiance_fused / hsi_cube against np.interp + the oracle's at-sensor radiance + ILS
    for random monochromatic axes (clipping the MAKO band list at either end), random knot sets (sparse, dense, not
    covering the axis), odd emissivity counts, resFactor and surface temperatures."""
    import torch
    from radtxfr_amd import engine, sensor

    rng = np.random.default_rng(20261014)
    dev = torch.device("cuda")
    f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
    for trial in range(6):
        lo = float(rng.uniform(740.0, 900.0))
        hi = float(rng.uniform(1100.0, 1340.0))
        n = int(rng.integers(20000, 60000))
        grid = engine.Grid(lo, hi, n)
        X = grid.axis()
        nk = int(rng.choice([2, 7, 60, 700]))
        Xk = np.sort(rng.uniform(lo - 30.0, hi + 30.0, nk)) if trial % 2 else np.sort(rng.uniform(lo + 40.0, hi - 40.0, nk))
        nE = int(rng.choice([1, 3, 5, 8]))
        em = rng.uniform(0.05, 1.0, (nk, nE))
        tau = 0.5 + 0.45 * np.sin(X / rng.uniform(5, 40))
        La = 2.0 + np.cos(X / rng.uniform(5, 40))
        Ld = 4.0 + 2.0 * np.sin(X / rng.uniform(3, 20))
        Ts = float(rng.uniform(260.0, 330.0))
        rf = None if trial % 3 else 2
        em_hi = np.stack([np.interp(X, Xk, em[:, k]) for k in range(nE)], axis=1)
        L_ref = ref.compute_LWIR_apparent_radiance(X, em_hi, np.array([Ts]), tau[:, None], La[:, None], Ld[:, None])[:, :, 0]
        xr, Lb_ref = ref.ILS_MAKO(X, L_ref, resFactor=rf)
        tag = (trial, lo, hi, n, nk, nE, Ts, rf)
        xo, Lb = sensor.band_radiance(grid, f32(tau), f32(La), f32(Ld), Xk, f32(em), Ts, resFactor=rf)
        assert np.array_equal(xo, xr) and rel_err(Lb.cpu().numpy(), Lb_ref) <= TOL_L, ("unfused",) + tag
        xf, Lf = sensor.band_radiance_fused(grid, f32(tau), f32(La), f32(Ld), Xk, f32(em), Ts, resFactor=rf)
        assert np.array_equal(xf, xr) and rel_err(Lf.cpu().numpy(), Lb_ref) <= TOL_L, ("fused",) + tag
        # pixel cube: mixtures of the nE columns with per-pixel temperatures
        n_pix = int(rng.integers(1, 70))
        n_mix = int(rng.integers(1, min(nE, 3) + 1))
        kidx = rng.integers(0, nE, (n_pix, n_mix)).astype(np.int32)
        frac = rng.uniform(0.0, 1.0, (n_pix, n_mix))
        frac = frac / frac.sum(1, keepdims=True)
        Tp = rng.uniform(270.0, 330.0, n_pix)
        xo, cube = sensor.hsi_cube(grid, f32(tau), f32(La), f32(Ld), Xk, f32(em), torch.as_tensor(kidx, device=dev), f32(frac),
                                   torch.as_tensor(Tp, device=dev), resFactor=rf)
        em_p = np.einsum("pm,xpm->xp", frac, em_hi[:, kidx])
        L = tau[:, None] * (em_p * ref.planckian(X, Tp) + (1 - em_p) * Ld[:, None]) + La[:, None]
        xr2, Lr = ref.ILS_MAKO(X, L, resFactor=rf)
        assert np.array_equal(xo, xr2) and cube.shape == Lr.reshape(xr2.size, -1).shape, ("cube",) + tag
        assert rel_err(cube.cpu().numpy(), Lr.reshape(xr2.size, -1)) <= TOL_L, ("cube",) + tag


def test_compute_tud_random_atmospheres_vs_oracle(rt):
    """The drop-in rt.compute_TUD end to end (prologue, line-sum, TUD) against the oracle for random atmospheres: layer
    counts 3..60 of the standard atmosphere (up to 120 km: Doppler-dominated lines, the fp64 pass), perturbed
    temperatures, mixing ratios scaled from opaque to thin, sensor altitudes inside the column, slant paths, grid steps."""
    rng = np.random.default_rng(20261015)
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    A = synthetic.load_standard_atmosphere()
    for trial in range(5):
        nL = int(rng.integers(3, 61))
        a = {"Zs": A[:nL, 1].copy(), "Ts": A[:nL, 5] + rng.uniform(-15.0, 15.0, nL), "Ps": A[:nL, 4].copy(), "PLs": A[:nL, 3].copy(),
             "MFs_VAL": A[:nL, 6:8] * 1e6 * 10.0 ** rng.uniform(-4.5, 0.0), "MFs_ID": np.array([1, 2])}
        dv = float(rng.choice([0.0005, 0.001, 0.004]))
        lo = float(rng.uniform(520.0, 5900.0))
        hi = lo + dv * int(rng.integers(1500, 4000))
        sub = synthetic.subset_table(full, lo - 12.0, hi + 12.0)
        alts = [500] if trial % 2 == 0 else sorted(rng.uniform(a["Zs"][0], a["Zs"][-1] + 5.0, 2).tolist())
        theta = float(rng.choice([0.0, 0.5]))
        nA = int(rng.choice([30, 12]))
        X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=dv, line_table=sub, Altitudes=np.asarray(alts), theta_r=theta, N_angle=nA, **a)
        Xr, tr, ur, dr = ref.compute_TUD(sub, lo, hi, dv, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"],
                                        Altitudes=alts, theta_r=theta, N_angle=nA)
        tag = (trial, nL, dv, lo, hi, alts, theta, nA)
        assert np.array_equal(X, Xr) and tau.shape == np.shape(tr) and Lu.shape == np.shape(ur), tag
        assert np.max(np.abs(tau - tr)) <= TOL_TAU, tag
        assert rel_err(Lu, ur) <= TOL_L and rel_err(Ld, dr) <= TOL_L, tag


def test_reduce_resolution_random_configurations_vs_oracle(rt):
    """rt.smooth / rt.reduceResolution against the oracle (NumPy convolution + scipy's cubic interp1d) for random axis
    lengths, window lengths (odd and even), window kinds, N, and explicit X_out inside the supported interior."""
    rng = np.random.default_rng(20261016)
    for trial in range(8):
        n = int(rng.integers(2000, 20000))
        lo = float(rng.uniform(500.0, 3000.0))
        h = float(10.0 ** rng.uniform(-3.5, -2.0))
        X = lo + h * np.arange(n)
        Y = 1.0 + 0.5 * np.sin(rng.uniform(20, 400) * X) + 0.05 * rng.standard_normal(n)
        nC = int(rng.choice([1, 2, 5]))
        Y2 = np.stack([Y * (k + 1) + k for k in range(nC)], axis=1)
        wl = int(rng.integers(3, 120))
        win = str(rng.choice(["flat", "hanning", "hamming", "bartlett", "blackman"]))
        assert rel_err(rt.smooth(Y, wl, win), ref.smooth(Y, wl, win)) < 1e-12, (trial, n, wl, win)
        sm = int(rng.integers(40, min(300, n // 6)))
        dX = sm * h
        N = int(rng.choice([2, 4, 8]))
        Xo_ref, Yo_ref = ref.reduceResolution(X, Y2 if nC > 1 else Y, dX, N=N, window=win)
        Xo, Yo = rt.reduceResolution(X, Y2 if nC > 1 else Y, dX, N=N, window=win)
        tag = (trial, n, h, sm, N, win, nC)
        if Xo.shape != Xo_ref.shape:  # the reference's own ceil() rounding flip: compare on its axis instead
            Yo = rt.reduceResolution(X, Y2 if nC > 1 else Y, dX, N=N, window=win, X_out=Xo_ref)
        else:
            assert np.allclose(Xo, Xo_ref, rtol=1e-12, atol=0), tag
        assert Yo.shape == Yo_ref.shape and rel_err(Yo, Yo_ref) < 1e-9, tag
        Xq = np.sort(rng.uniform(X[sm // 2 + 24], X[-sm // 2 - 26], 57))
        assert rel_err(rt.reduceResolution(X, Y, dX, N=N, window=win, X_out=Xq), ref.reduceResolution(X, Y, dX, N=N, window=win, X_out=Xq)) < 1e-9, tag
    # a short window: the first output points lie among the distorted end knots -- evaluated by the local end-region spline
    xs_r, ys_r = ref.reduceResolution(X, Y, 12 * h)
    ys = rt.reduceResolution(X, Y, 12 * h, X_out=xs_r)  # (on the reference's axis: its point count can flip by one)
    assert ys.shape == ys_r.shape and rel_err(ys, ys_r) < 1e-9


def test_hapi_shim_option_combinations_vs_oracle(hapi):
    """The option surface of hapi.absorptionCoefficient_Voigt (misc/hapi.py:10906-11141) in random combinations: explicit
    Components with custom abundances / a filtered-out molecule, HITRAN_units on/off, GammaL air/self, Diluent mixes,
    IntensityThreshold > 0, OmegaRange + OmegaStep instead of a grid, two SourceTables at once."""
    rng = np.random.default_rng(20261018)
    full = synthetic.synth_line_table(synthetic.SEED_C2, 3000, 675.0, 1425.0)
    for trial in range(8):
        lo = float(rng.uniform(700.0, 1350.0))
        step = float(rng.choice([0.002, 0.005, 0.01]))
        n = int(rng.integers(800, 5000))
        hi = lo + step * n
        sub = synthetic.subset_table(full, lo - 8.0, hi + 8.0)
        nsub = len(sub["nu"])
        # split the lines over two cached tables (the shim concatenates SourceTables)
        cut = nsub // 3
        t1 = {k: np.asarray(v)[:cut] for k, v in sub.items()}
        t2 = {k: np.asarray(v)[cut:] for k, v in sub.items()}
        hapi.storage2cache_from_columns("opt_a", t1)
        hapi.storage2cache_from_columns("opt_b", t2)
        Tk, p = float(rng.uniform(200.0, 320.0)), float(10.0 ** rng.uniform(-2.0, 0.05))
        comps = [None, [(1, 1)], [(1, 1), (2, 1, 0.37)], [(2, 1, 0.011)]][trial % 4]
        units = bool(trial % 2)
        gl = "gamma_self" if trial % 3 == 1 else "gamma_air"
        dil = {"air": 0.55, "self": 0.45} if trial % 3 == 2 else {}
        thr = float(np.percentile(sub["sw"], 40)) if trial % 4 == 3 else 0.0
        kw = dict(Components=comps, SourceTables=["opt_a", "opt_b"], Environment={"T": Tk, "p": p}, HITRAN_units=units, GammaL=gl,
                  Diluent=dil, IntensityThreshold=thr, OmegaWingHW=25.0)
        if trial % 2:
            om, xs = hapi.absorptionCoefficient_Voigt(OmegaRange=[lo, hi], OmegaStep=step, **kw)
        else:
            om, xs = hapi.absorptionCoefficient_Voigt(OmegaGrid=np.linspace(lo, hi, n), **kw)
        _, want = ref.absorptionCoefficient_Voigt(sub, Components=comps, T=Tk, p=p, OmegaGrid=om, OmegaWingHW=25.0, HITRAN_units=units,
                                                  GammaL=gl, Diluent=dil or None, IntensityThreshold=thr)
        tag = (trial, lo, step, n, Tk, p, comps, units, gl, dil, thr)
        assert np.max(want) > 0 and rel_err(xs, want) <= TOL_L, tag


def test_drop_in_edge_cases(rt, hapi, tmp_path):
    """Degenerate but legal calls through the drop-in API: an empty line table, one line on a 4-point grid, one layer
    with one angle (the reference's 0/0), a sensor below the column, returnOD with save=True (ComputeTUD.npz), a grid no
    line reaches, a 2-point grid."""
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    a = synthetic.c3_atmosphere(32)
    args = (a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    empty = {k: np.asarray(v)[:0] for k, v in full.items()}
    X, tau, Lu, Ld = rt.compute_TUD(1000.0, 1000.5, DVOUT=0.001, line_table=empty, **a)
    assert X.shape == (500,) and np.all(tau == 1.0) and np.all(Lu == 0.0) and np.all(Ld == 0.0)
    one = {k: np.asarray(v)[5000:5001] for k, v in full.items()}
    nu0 = float(one["nu"][0])
    X, tau, Lu, Ld = rt.compute_TUD(nu0 - 0.002, nu0 + 0.002, DVOUT=0.001, line_table=one, **a)
    Xr, tr, ur, dr = ref.compute_TUD(one, nu0 - 0.002, nu0 + 0.002, 0.001, *args)
    assert X.shape == (4,) and np.max(np.abs(tau - tr)) <= TOL_TAU and rel_err(Lu, ur) <= TOL_L and rel_err(Ld, dr) <= TOL_L
    sub = synthetic.subset_table(full, 988.0, 1016.0)
    a1 = {k: (np.asarray(v)[:1] if k != "MFs_ID" else v) for k, v in a.items()}
    X, tau, Lu, Ld = rt.compute_TUD(1000.0, 1001.0, DVOUT=0.001, line_table=sub, N_angle=1, **a1)
    assert np.isnan(Ld).all() and np.isfinite(tau).all() and np.isfinite(Lu).all()
    X, tau, Lu, Ld = rt.compute_TUD(1000.0, 1001.0, DVOUT=0.001, line_table=sub, Altitudes=np.asarray([-5.0]), **a)
    Xr, tr, ur, dr = ref.compute_TUD(sub, 1000.0, 1001.0, 0.001, *args, Altitudes=[-5.0])
    assert np.max(np.abs(tau - tr)) <= TOL_TAU and np.max(np.abs(Lu - ur)) == 0.0 and np.array_equal(np.isnan(Ld), np.isnan(dr))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        X, od, Lu, Ld = rt.compute_TUD(1000.0, 1001.0, DVOUT=0.001, line_table=sub, returnOD=True, save=True, **a)
    finally:
        os.chdir(cwd)
    Xr, odr, ur, dr = ref.compute_TUD(sub, 1000.0, 1001.0, 0.001, *args, returnOD=True)
    assert rel_err(od, odr) <= TOL_L and rel_err(Lu, ur) <= TOL_L
    z = np.load(os.path.join(tmp_path, "ComputeTUD.npz"))
    assert sorted(z.files) == ["B", "Ld", "Lu", "OD", "X", "Z_s", "angles", "mu_s", "tau"] and z["OD"].shape == (X.size, 32)
    hapi.storage2cache_from_columns("edge", sub)
    om, xs = hapi.absorptionCoefficient_Voigt(SourceTables="edge", OmegaGrid=np.linspace(3000.0, 3001.0, 11))
    assert xs.shape == (11,) and np.all(xs == 0.0)
    om, xs = hapi.absorptionCoefficient_Voigt(SourceTables="edge", OmegaGrid=np.array([1000.0, 1000.5]))
    _, xr = ref.absorptionCoefficient_Voigt(sub, OmegaGrid=np.array([1000.0, 1000.5]))
    assert rel_err(xs, xr) <= TOL_L


# ----------------------------------------------------------------------- round 2: .par reader -> device, slant vectors
def test_g12_par_files_to_device_line_sum(hapi, golden):
    """SURVEY 8f row 3 end to end: .par / .data + .header files -> hapi.db_begin (the reference's storage2cache rules,
    radtxfr_amd/hitran_par.py) -> device line table -> absorptionCoefficient_Voigt, against what the REFERENCE computed
    from its own parse of the same files (golden G12): air broadening on the header-less .par (isotopologue code '0'
    -> (2, 0)), air + self with the n_self / deltap_air / delta_self / deltap_self extras on the .data table."""
    from conftest import GOLDEN
    g = golden("g12_par_tables.npz")
    names = hapi.db_begin(GOLDEN)
    assert set(names) == {"g12a", "g12b"}
    grid = np.linspace(float(g["grid_lo"]), float(g["grid_hi"]), int(g["grid_n"]))
    env = {"T": float(g["T"]), "p": float(g["p"])}
    _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="g12a", Environment=env, OmegaGrid=grid, HITRAN_units=True)
    assert rel_err(xs, g["g12a_xs"]) <= TOL_L
    _, xs = hapi.absorptionCoefficient_Voigt(SourceTables="g12b", Environment=env, OmegaGrid=grid, HITRAN_units=True,
                                             Diluent={"air": 0.6, "self": 0.4})
    assert rel_err(xs, g["g12b_xs"]) <= TOL_L
    # without the optional columns the answer differs visibly: they are really used
    d = hapi.LOCAL_TABLE_CACHE["g12b"]["data"]
    hapi.LOCAL_TABLE_CACHE["g12b_plain"] = {"header": {"number_of_rows": len(d["nu"])},
                                            "data": {k: v for k, v in d.items() if k not in ("n_self", "deltap_air", "delta_self", "deltap_self")}}
    _, xs_plain = hapi.absorptionCoefficient_Voigt(SourceTables="g12b_plain", Environment=env, OmegaGrid=grid,
                                                   Diluent={"air": 0.6, "self": 0.4})
    assert rel_err(xs_plain, g["g12b_xs"]) > 1e-3
    for n in ("g12a", "g12b", "g12b_plain"):
        hapi.LOCAL_TABLE_CACHE.pop(n)


def test_cached_device_table_sees_in_place_edits(hapi):
    """the reference re-reads LOCAL_TABLE_CACHE on every call (misc/hapi.py:11044-11125); the device copy
    here must not survive an in-place edit of ANY column."""
    tbl = synthetic.synth_line_table(11, 300, 995.0, 1005.0)
    hapi.LOCAL_TABLE_CACHE["edit"] = {"header": {"number_of_rows": 300}, "data": {k: np.array(v) for k, v in tbl.items()}}
    grid = np.linspace(998.0, 1002.0, 4001)
    kw = dict(SourceTables="edit", Environment={"T": 280.0, "p": 0.8}, OmegaGrid=grid)
    _, x0 = hapi.absorptionCoefficient_Voigt(**kw)
    hapi.LOCAL_TABLE_CACHE["edit"]["data"]["sw"] *= 2.0                      # same array object, same row count
    _, x1 = hapi.absorptionCoefficient_Voigt(**kw)
    assert rel_err(x1, 2.0 * x0) <= 1e-6
    hapi.LOCAL_TABLE_CACHE["edit"]["data"]["gamma_air"][:] = 0.5 * hapi.LOCAL_TABLE_CACHE["edit"]["data"]["gamma_air"]
    _, x2 = hapi.absorptionCoefficient_Voigt(**kw)
    t2 = dict(tbl, sw=tbl["sw"] * 2.0, gamma_air=tbl["gamma_air"] * 0.5)
    _, x2r = ref.absorptionCoefficient_Voigt(t2, T=280.0, p=0.8, OmegaGrid=grid)
    assert rel_err(x2, x2r) <= TOL_L and rel_err(x2, x1) > 1e-2
    hapi.LOCAL_TABLE_CACHE.pop("edit")


def test_g13_compute_tud_vector_theta_golden(rt, golden):
    """Vector theta_r through the drop-in (radiative_transfer.py:313, 346-365): (nX, nZ, nMu) for 2 altitudes x 2
    slants, (nX, nMu) for one altitude x 3 slants -- the reference's squeeze rules -- against golden G13; and more
    slant paths than one rtx_tud launch takes (9 > 8) against the oracle."""
    g = golden("g13_tud_slants.npz")
    full = synthetic.synth_line_table(int(g["seed"]), int(g["n_lines"]), float(g["nu_lo"]), float(g["nu_hi"]))
    lo, hi = float(g["lo"]), float(g["hi"])
    sub = synthetic.subset_table(full, lo - float(g["pad"]), hi + float(g["pad"]))
    a = synthetic.c3_atmosphere(32)
    a["MFs_VAL"] = a["MFs_VAL"] * float(g["mf_scale"])
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, theta_r=g["th22"], Altitudes=g["alt22"], **a)
    assert tau.shape == Lu.shape == (X.size, 2, 2) and Ld.shape == (X.size,)
    assert np.max(np.abs(tau - g["tau22"])) <= TOL_TAU
    assert rel_err(Lu, g["Lu22"]) <= TOL_L and rel_err(Ld, g["Ld22"]) <= TOL_L
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, theta_r=g["th13"], Altitudes=np.asarray([500]), **a)
    assert tau.shape == Lu.shape == (X.size, 3)
    assert np.max(np.abs(tau - g["tau13"])) <= TOL_TAU
    assert rel_err(Lu, g["Lu13"]) <= TOL_L and rel_err(Ld, g["Ld13"]) <= TOL_L
    th9 = np.linspace(0.0, 1.2, 9)
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, theta_r=th9, Altitudes=np.asarray([3.0, 9.0]), **a)
    Xr, tau_r, Lu_r, Ld_r = ref.compute_TUD(sub, lo, hi, 0.001, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"],
                                            Altitudes=[3.0, 9.0], theta_r=th9)
    assert tau.shape == tau_r.shape == (X.size, 2, 9)
    assert np.max(np.abs(tau - tau_r)) <= TOL_TAU and rel_err(Lu, Lu_r) <= TOL_L and rel_err(Ld, Ld_r) <= TOL_L


def test_compute_tud_batch_equals_per_call_results(rt):
    """compute_TUD_batch (the reference's loop over atmospheres, Generate_LWIR_TUD.py:117-150, as a device pipeline: the
    copy of atmosphere k overlaps the kernels of k+1, double-buffered outputs) returns exactly what one compute_TUD call
    per atmosphere returns; with reduce= it returns what reduceResolution makes of those (Generate_LWIR_TUD.py:124-126)."""
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi = 1000.0, 1004.0
    sub = synthetic.subset_table(full, lo - 12.0, hi + 12.0)
    a = synthetic.c3_atmosphere(32)
    a["MFs_VAL"] = a["MFs_VAL"] * 1e-3
    rng = np.random.default_rng(3)
    atms = [dict(Ts=a["Ts"] + rng.normal(0, 2.0, 32), MFs_VAL=a["MFs_VAL"] * rng.uniform(0.5, 1.5, (32, 1)), Ps=a["Ps"] * s_)
            for s_ in (1.0, 0.98, 1.01, 1.0, 0.95)]
    common = dict(DVOUT=0.001, line_table=sub, Zs=a["Zs"], PLs=a["PLs"], MFs_ID=a["MFs_ID"], Ts=a["Ts"], Ps=a["Ps"],
                  MFs_VAL=a["MFs_VAL"], Altitudes=np.asarray([2.0, 500.0]), theta_r=0.3)
    got = rt.compute_TUD_batch(lo, hi, atms, **common)
    assert len(got) == len(atms)
    for g, atm in zip(got, atms):
        X, tau, Lu, Ld = rt.compute_TUD(lo, hi, **dict(common, **atm))
        assert np.array_equal(g[0], X) and g[1].shape == tau.shape == (X.size, 2)
        assert np.array_equal(g[1], tau) and np.array_equal(g[2], Lu) and np.array_equal(g[3], Ld)
    assert not np.array_equal(got[0][1], got[1][1])
    # oracle on one of them
    k = 2
    at = dict(common, **atms[k])
    Xr, tau_r, Lu_r, Ld_r = ref.compute_TUD(sub, lo, hi, 0.001, at["Zs"], at["Ts"], at["Ps"], at["PLs"], at["MFs_VAL"], at["MFs_ID"],
                                            Altitudes=[2.0, 500.0], theta_r=0.3)
    assert np.max(np.abs(got[k][1] - tau_r)) <= TOL_TAU and rel_err(got[k][2], Lu_r) <= TOL_L and rel_err(got[k][3], Ld_r) <= TOL_L
    # reduced on the device
    red = rt.compute_TUD_batch(lo, hi, atms[:3], reduce=dict(dX=0.25), **common)
    for g, full_res in zip(red, got):
        Xo, t_r = rt.reduceResolution(full_res[0], full_res[1], 0.25)
        assert np.array_equal(g[0], Xo) and g[1].shape == t_r.shape
        assert rel_err(g[1], t_r) <= 1e-12 and rel_err(g[2], rt.reduceResolution(full_res[0], full_res[2], 0.25, X_out=Xo)) <= 1e-12
        assert rel_err(g[3], rt.reduceResolution(full_res[0], full_res[3], 0.25, X_out=Xo)) <= 1e-12


def test_compute_tud_batch_devices_and_host_paths(rt):
    """compute_TUD_batch(devices=[0, 0]): two independent pipelines (own streams, own per-(line, layer) records, one
    shared line table per device) dealt the atmospheres round-robin -- the reference's multiprocessing.Pool axis
    (Generate_LWIR_TUD.py:117-150) from one process -- return, in input order, exactly what one pipeline returns;
    out_dtype=float32 is the same numbers unwidened; a device-resident LineTable is accepted; and the zero-copy pinned
    results of compute_TUD fall back to pageable arrays, with identical values, once the pinned cap is lent out."""
    import torch
    from radtxfr_amd import _hostio, engine
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi = 1000.0, 1004.0
    sub = synthetic.subset_table(full, lo - 12.0, hi + 12.0)
    a = synthetic.c3_atmosphere(32)
    a["MFs_VAL"] = a["MFs_VAL"] * 1e-3
    rng = np.random.default_rng(5)
    atms = [dict(Ts=a["Ts"] + rng.normal(0, 2.0, 32), MFs_VAL=a["MFs_VAL"] * rng.uniform(0.5, 1.5, (32, 1))) for _ in range(7)]
    lines = engine.LineTable(sub)
    common = dict(DVOUT=0.001, line_table=lines, Zs=a["Zs"], PLs=a["PLs"], MFs_ID=a["MFs_ID"], Ts=a["Ts"], Ps=a["Ps"],
                  MFs_VAL=a["MFs_VAL"], Altitudes=np.asarray([500.0]))
    one = rt.compute_TUD_batch(lo, hi, atms, **common)
    two = rt.compute_TUD_batch(lo, hi, atms, devices=[0, 0], **common)
    f32 = rt.compute_TUD_batch(lo, hi, atms, devices=[0, 0], out_dtype=np.float32, **common)
    assert len(one) == len(two) == len(f32) == 7
    for r1, r2, r3 in zip(one, two, f32):
        assert r1[1].dtype == np.float64 and r3[1].dtype == np.float32 and r1[1].flags.writeable
        for j in (1, 2, 3):
            assert np.array_equal(r1[j], r2[j]) and np.array_equal(r1[j].astype(np.float32), r3[j])
    assert not np.array_equal(one[0][1], one[1][1])
    with pytest.raises(ValueError):
        rt.compute_TUD_batch(lo, hi, atms, devices=[torch.cuda.device_count()], **common)
    # zero-copy results are page-locked while they live; past the cap compute_TUD hands out pageable arrays
    base = _hostio.pinned_lent_bytes()
    r_pin = rt.compute_TUD(lo, hi, **dict(common, **atms[0]))
    per_result = _hostio.pinned_lent_bytes() - base
    assert per_result == 3 * r_pin[1].size * 8
    cap0 = _hostio.PINNED_RESULT_CAP
    try:
        _hostio.PINNED_RESULT_CAP = _hostio.pinned_lent_bytes() + per_result // 2
        r_page = rt.compute_TUD(lo, hi, **dict(common, **atms[0]))
        assert _hostio.pinned_lent_bytes() == base + per_result  # nothing more was lent out
    finally:
        _hostio.PINNED_RESULT_CAP = cap0
    for j in (1, 2, 3):
        assert np.array_equal(r_pin[j], r_page[j]) and np.array_equal(r_pin[j], one[0][j])
    del r_pin
    import gc
    gc.collect()
    assert _hostio.pinned_lent_bytes() == base
    # the axis: cached and read-only by default, a fresh writable copy on request (the reference's behaviour)
    Xc = rt.compute_TUD(lo, hi, **dict(common, **atms[0]))[0]
    Xw = rt.compute_TUD(lo, hi, copy_axis=True, **dict(common, **atms[0]))[0]
    assert not Xc.flags.writeable and Xw.flags.writeable and np.array_equal(Xc, Xw) and Xw is not Xc
    Xw *= 2.0
    assert np.array_equal(rt.compute_TUD(lo, hi, **dict(common, **atms[0]))[0], Xc)
    lines.close()


def test_reduce_resolution_end_regions_vs_oracle(rt):
    """reduceResolution with SHORT windows and output points close to the ends of the axis (radiative_transfer.py:1327-1350):
    there the reference's spline runs on the smoothed -- no longer uniform -- axis and its not-a-knot end condition acts; the
    engine evaluates those points with a local not-a-knot spline on the true knots (rtx_cubic_end), the rest with the
    cardinal spline. Against the oracle (scipy's interp1d through all smoothed samples): default X_out for windows of 5 to
    60 samples, explicit X_out reaching to the first and last smoothed knot, 2-D Y, float32 device rows."""
    import torch
    from radtxfr_amd import engine
    rng = np.random.default_rng(21)
    n = 6000
    X = np.linspace(700.0, 760.0 - 0.01, n)  # step 0.01
    Y = np.exp(-0.5 * ((X[:, None] - np.array([700.3, 730.0, 759.6])) / np.array([0.4, 2.5, 0.3])) ** 2) + 0.05 * rng.standard_normal((n, 3)) * 0.01
    for dX in (0.05, 0.1, 0.21, 0.37, 0.6):
        xo_r, yo_r = ref.reduceResolution(X, Y, dX)
        xo, yo = rt.reduceResolution(X, Y, dX)
        assert xo.shape == xo_r.shape and np.allclose(xo, xo_r, rtol=0, atol=1e-9), dX
        assert rel_err(yo, yo_r) <= 1e-9, (dX, rel_err(yo, yo_r))
        y1 = rt.reduceResolution(X, Y[:, 1], dX, X_out=xo)
        assert rel_err(y1, yo_r[:, 1]) <= 1e-9
    # explicit abscissae from the very first to the very last smoothed knot
    dX = 0.1
    Xs = ref.smooth_sym(X, 10)
    xq = np.concatenate([np.linspace(Xs[0], Xs[40], 57), np.linspace(X[100], X[-100], 301), np.linspace(Xs[-41], Xs[-1], 63)])
    want = ref.reduceResolution(X, Y, dX, X_out=xq)
    got = rt.reduceResolution(X, Y, dX, X_out=xq)
    assert rel_err(got, want) <= 1e-9, rel_err(got, want)
    with pytest.raises(NotImplementedError):
        rt.reduceResolution(X, Y, dX, X_out=np.array([X[0] - 1.0, X[50]]))  # extrapolation
    # device-resident float32 rows through the engine (what compute_TUD_batch(reduce=...) does), short window
    rows = torch.as_tensor(Y.T.astype(np.float32), device="cuda").contiguous()
    xo_e, out = engine.reduce_resolution_cached(rows, float(X[0]), float(X[1] - X[0]), n, 0.1)
    xo_r, yo_r = ref.reduceResolution(X, Y.astype(np.float32).astype(np.float64), 0.1)
    assert np.allclose(xo_e, xo_r, rtol=0, atol=1e-9) and rel_err(out.cpu().numpy().T, yo_r) <= 1e-9


def test_compute_tud_chunked_equals_unchunked(rt):
    """compute_TUD in tile-aligned wavenumber chunks (each chunk's device-to-host copy under the next chunk's kernels: a
    single call is PCIe-bound) returns the unchunked arrays bit for bit -- one and several (altitude, slant) pairs, more
    chunks than tiles, returnOD -- and the automatic choice (4 chunks from 2 M points) equals chunks=1 on a C3-sized call."""
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi = 1000.0, 1004.5
    sub = synthetic.subset_table(full, lo - 12.0, hi + 12.0)
    a = synthetic.c3_atmosphere(32)
    a["MFs_VAL"] = a["MFs_VAL"] * 1e-3
    for extra in (dict(), dict(Altitudes=np.asarray([2.0, 500.0]), theta_r=np.asarray([0.0, 0.7])), dict(returnOD=True)):
        base = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, chunks=1, **extra, **a)
        for k in (2, 3, 9):
            got = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, chunks=k, **extra, **a)
            assert all(g.shape == b.shape and np.array_equal(g, b) for g, b in zip(got, base)), (extra.keys(), k)
    big1 = rt.compute_TUD(500.0, 6000.0, DVOUT=0.002, line_table=full, chunks=1, **a)
    big4 = rt.compute_TUD(500.0, 6000.0, DVOUT=0.002, line_table=full, **a)  # 2.75 M points: automatic = 4 chunks
    assert big1[1].size == 2750000 and all(np.array_equal(g, b) for g, b in zip(big4, big1))
    Xr, tau_r, Lu_r, Ld_r = ref.compute_TUD(sub, lo, hi, 0.001, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, chunks=3, **a)
    assert np.max(np.abs(tau - tau_r)) <= TOL_TAU and rel_err(Lu, Lu_r) <= TOL_L and rel_err(Ld, Ld_r) <= TOL_L


def test_compute_tud_save_with_many_slants(rt, tmp_path, monkeypatch):
    """save=True (radiative_transfer.py:374-386) with more slant paths than one rtx_tud launch takes: the slants run in
    blocks and the per-stream downwelling radiances -- independent of the slant -- come from the first block."""
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi = 1000.0, 1001.0
    sub = synthetic.subset_table(full, lo - 12.0, hi + 12.0)
    a = synthetic.c3_atmosphere(8)
    a["MFs_VAL"] = a["MFs_VAL"] * 1e-3
    monkeypatch.chdir(tmp_path)
    th = np.linspace(0.0, 1.1, 10)
    X, tau, Lu, Ld = rt.compute_TUD(lo, hi, DVOUT=0.001, line_table=sub, theta_r=th, save=True, N_angle=6, **a)
    d = np.load(tmp_path / "ComputeTUD.npz")
    assert tau.shape == (X.size, 10) and d["Ld"].shape == (X.size, 6) and d["tau"].shape == (X.size, 1, 10)
    Xr, tau_r, Lu_r, Ld_r = ref.compute_TUD(sub, lo, hi, 0.001, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"],
                                            theta_r=th, N_angle=6)
    assert np.max(np.abs(tau - tau_r)) <= TOL_TAU and rel_err(Lu, Lu_r) <= TOL_L and rel_err(Ld, Ld_r) <= TOL_L
    ang = np.linspace(0, np.pi / 2.0, 6, endpoint=False)
    w = np.cos(ang) * np.sin(ang)
    assert rel_err((d["Ld"] * w).sum(axis=1) / w.sum(), Ld_r) <= TOL_L


def test_fused_entry_equals_three_calls():
    """rtx_compute_tud (engine.TudRunner: one library call per atmosphere, per-layer tables in the prologue's kernel
    arguments) gives bit-identical tau / L-up / L-down / OD to rtx_line_prep + rtx_voigt_sum + rtx_tud; also with more
    species x layers than the kernel-argument block holds (the device-buffer path)."""
    import torch
    from radtxfr_amd import engine
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    sub = synthetic.subset_table(full, 2340.0, 2362.0)
    grid = engine.Grid(2350.0, 2352.0, 2000)
    for nlay in (32, 66):
        A = synthetic.load_standard_atmosphere()[:nlay]
        a = dict(Zs=A[:, 1], Ts=A[:, 5], Ps=A[:, 4], PLs=A[:, 3], MFs_VAL=A[:, 6:8] * 1e6 * 1e-2, MFs_ID=np.array([1, 2]))
        lines = engine.LineTable(sub)
        assert (2 + 2 * len(lines.species)) * nlay + len(lines.species) > 416 or nlay == 32
        OD = engine.optical_depths(lines, grid, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
        tau, Lu, Ld, _ = engine.tud(OD, grid, a["Ts"], a["Zs"], Altitudes=[3.0, 500.0], theta_r=0.4)
        run = engine.TudRunner(lines, grid, a["Zs"], n_layers=nlay, Altitudes=[3.0, 500.0], theta_r=0.4)
        t2, u2, d2 = run.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
        torch.cuda.synchronize()
        assert torch.equal(run.OD, OD) and torch.equal(t2, tau) and torch.equal(u2, Lu) and torch.equal(d2, Ld)
        lines.close()


# ----------------------------------------------------------------------- C4 / C5 at FULL size, sampled against the oracle
def _mako_span_tud():
    """tau, L-up, L-down of the engine on the MAKO span of the C3 grid (755-1325 cm^-1 at 0.001 cm^-1: 570 000 points, a
    shard of the 5.5 M-point axis), 32 layers, mixing ratios x 3e-3 so that the span has windows and bands."""
    import torch
    from radtxfr_amd import engine
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    g_full = engine.Grid(500.0, 6000.0, 5500000)
    i0, i1 = int((755.0 - 500.0) / g_full.step), int((1325.0 - 500.0) / g_full.step)
    grid = g_full.shard(i0, i1 - i0)
    a = synthetic.c3_atmosphere(32)
    lines = engine.LineTable(synthetic.subset_table(full, 740.0, 1340.0))
    run = engine.TudRunner(lines, grid, a["Zs"], n_layers=32)
    tau, Lu, Ld = run.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"] * 3e-3, a["MFs_ID"])
    torch.cuda.synchronize()
    out = grid, tau[0].clone(), Lu[0].clone(), Ld.clone()
    lines.close()
    return out


def test_c4_full_size_sampled_columns_vs_oracle(rt):
    """Config C4 at BASELINE.json's full size -- 2000 emissivities x 570 000 monochromatic points x 128 MAKO bands -- on the
    engine's own TUD of that span: 24 of the 2000 columns are checked against the oracle evaluated at the full spectral
    size (np.interp of the knots, compute_LWIR_apparent_radiance, the reference's ILS); the streaming pipeline and the fused
    one must both match, and agree with each other on all 2000 columns."""
    import torch
    from radtxfr_amd import sensor
    grid, tau, La, Ld = _mako_span_tud()
    X = grid.axis()
    t64, a64, d64 = (v.double().cpu().numpy() for v in (tau, La, Ld))
    assert 0.02 < np.median(t64) < 0.999 and t64.min() < 0.05 and t64.max() > 0.5
    Xe, em = synthetic.synth_emissivities(n_emis=2000)
    em_d = torch.as_tensor(em.astype(np.float32), device="cuda")
    xo, Lb = sensor.band_radiance(grid, tau, La, Ld, Xe, em_d, 287.87)
    xf, Lf = sensor.band_radiance_fused(grid, tau, La, Ld, Xe, em_d, 287.87)
    assert Lb.shape == Lf.shape == (xo.size, 2000) and xo.size >= 126 and np.array_equal(xo, xf)
    assert rel_err(Lf.cpu().numpy(), Lb.cpu().numpy()) <= 3e-6
    cols = np.arange(7, 2000, 83)
    em_ref = np.stack([np.interp(X, Xe, em[:, k].astype(np.float32).astype(np.float64)) for k in cols], axis=1)
    L_ref = ref.compute_LWIR_apparent_radiance(X, em_ref, np.array([287.87]), t64[:, None], a64[:, None], d64[:, None])[:, :, 0]
    xr, Lb_ref = ref.ILS_MAKO(X, L_ref)
    assert np.array_equal(xo, xr)
    assert rel_err(Lb.cpu().numpy()[:, cols], Lb_ref) <= TOL_L
    assert rel_err(Lf.cpu().numpy()[:, cols], Lb_ref) <= TOL_L


def test_c5_full_size_sampled_pixels_vs_oracle(rt):
    """Config C5 at full size -- 256 x 256 pixels x 256 MAKO bands (resFactor 2) from 570 000 monochromatic points -- on the
    engine's own TUD of the span: 64 of the 65 536 pixels are checked against the oracle, which forms each pixel's
    monochromatic spectrum (its emissivity mixture, its surface temperature; LWIR_HSI_Generator.py:151-167) and runs the
    reference's ILS on it at the full spectral size."""
    import torch
    from radtxfr_amd import sensor
    grid, tau, La, Ld = _mako_span_tud()
    X = grid.axis()
    t64, a64, d64 = (v.double().cpu().numpy() for v in (tau, La, Ld))
    Xe, em = synthetic.synth_emissivities(n_emis=2000)
    sc = synthetic.synth_scene()  # 65 536 pixels, 2-of-6 mixtures, T = 287.87 + 3 N(0,1)
    E = em[:, sc["end_idx"]].astype(np.float32)
    dev = torch.device("cuda")
    xo, cube = sensor.hsi_cube(grid, tau, La, Ld, Xe, torch.as_tensor(E, device=dev), torch.as_tensor(sc["kidx"], device=dev),
                               torch.as_tensor(sc["frac"].astype(np.float32), device=dev), torch.as_tensor(sc["T"], device=dev),
                               resFactor=2)
    assert cube.shape == (xo.size, 256 * 256) and xo.size >= 252 and bool(torch.isfinite(cube).all())
    pix = np.arange(11, 65536, 1040)[:64]
    E_hi = np.stack([np.interp(X, Xe, E[:, k].astype(np.float64)) for k in range(E.shape[1])], axis=1)   # [nX][6]
    frac = sc["frac"].astype(np.float32).astype(np.float64)[pix]
    em_p = np.einsum("pm,xpm->xp", frac, E_hi[:, sc["kidx"][pix]])                                      # [nX][64]
    L = t64[:, None] * (em_p * ref.planckian(X, sc["T"][pix]) + (1 - em_p) * d64[:, None]) + a64[:, None]
    xr, Lr = ref.ILS_MAKO(X, L, resFactor=2)
    assert np.array_equal(xo, xr)
    assert rel_err(cube.cpu().numpy()[:, pix], Lr) <= TOL_L


def test_c5_full_size_end_to_end_vs_oracle(rt):
    """Config C5 at full size, END TO END: line table + atmosphere -> dist.hsi_cube_from_atmosphere (prologue, line-sum, TUD
    on the 570 000-point MAKO span, band moments, 256 x 256-pixel x 256-band cube). 16 sampled pixels are checked against
    an oracle that starts from the LINE TABLE too -- oracle line-sum and oracle TUD on a 20 000-point sub-span, each
    pixel's monochromatic spectrum, the reference's triangle ILS -- for every band whose triangle lies inside the
    sub-span, so the sensor stage is not only checked on the engine's own tau."""
    import torch
    from radtxfr_amd import dist as rdist
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi, dv = 755.0, 1325.0, 0.001
    sub = synthetic.subset_table(full, lo - 15.0, hi + 15.0)
    a = synthetic.c3_atmosphere(32)
    a["MFs_VAL"] = a["MFs_VAL"] * 3e-3
    Xe, em = synthetic.synth_emissivities(n_emis=2000)
    sc = synthetic.synth_scene()
    E = em[:, sc["end_idx"]].astype(np.float32)
    dev = torch.device("cuda")
    xo, cube = rdist.hsi_cube_from_atmosphere(lo, hi, dv, sub, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"], Xe,
                                              torch.as_tensor(E, device=dev), torch.as_tensor(sc["kidx"], device=dev),
                                              torch.as_tensor(sc["frac"].astype(np.float32), device=dev),
                                              torch.as_tensor(sc["T"], device=dev), resFactor=2)
    torch.cuda.synchronize()
    assert cube.shape == (xo.size, 256 * 256) and xo.size >= 252 and bool(torch.isfinite(cube).all())
    nX = int(np.ceil((hi - lo) / dv))
    X = np.linspace(lo, hi, nX)
    i0, n = 250000, 20000  # ~1005-1025 cm^-1
    Xw = X[i0:i0 + n]
    subw = synthetic.subset_table(full, Xw[0] - 12.0, Xw[-1] + 12.0)
    ODr = np.stack([ref.layer_od(subw, Xw, a["Ts"][k], a["Ps"][k], a["PLs"][k], a["MFs_VAL"][k], a["MFs_ID"]) for k in range(32)], 1)
    tr, ur, dr = ref.tud_from_od(Xw, ODr, a["Ts"], a["Zs"])
    assert tr.max() - tr.min() > 0.3
    pix = np.arange(5, 65536, 4099)[:16]
    E_hi = np.stack([np.interp(Xw, Xe, E[:, k].astype(np.float64)) for k in range(E.shape[1])], axis=1)
    frac = sc["frac"].astype(np.float32).astype(np.float64)[pix]
    em_p = np.einsum("pm,xpm->xp", frac, E_hi[:, sc["kidx"][pix]])
    Lw = tr[:, None] * (em_p * ref.planckian(Xw, sc["T"][pix]) + (1 - em_p) * dr[:, None]) + ur[:, None]
    L_full = np.zeros((nX, pix.size))
    L_full[i0:i0 + n] = Lw
    xr, Lr = ref.ILS_MAKO(X, L_full, resFactor=2)
    assert np.array_equal(xo, xr)
    sig = np.abs(np.gradient(xr)) * 1.6
    inside = (xr - sig > Xw[0]) & (xr + sig < Xw[-1])
    assert inside.sum() >= 3
    got = cube.cpu().numpy()[:, pix][inside]
    assert rel_err(got, Lr[inside]) <= TOL_L, rel_err(got, Lr[inside])


def test_c5_cube_kernel_paths_vs_oracle(rt):
    """The paths of the round-3 cube kernels the C5 configuration does not reach (LWIR_HSI_Generator.py:151-167 + rt.ILS_MAKO):
    a pixel count that ends inside a workgroup and inside a round of 64; mixtures of 6 endmembers (read per pixel, not staged
    in the lanes) and of 3 and 4; 400 endmembers (tables too large for LDS: read from global memory) and 31 (the largest
    tables that still go to LDS); Q = 3, 5, 6; pixels at
    1200-1500 K, where the Planck exponent of a MAKO band drops below 1.5 and the fp64 small-argument branch runs (decided
    per workgroup, then per pixel); a knot axis denser than the grid's bands (rounds of more than 16 intervals per band)."""
    import torch
    from radtxfr_amd import engine, sensor
    rng = np.random.default_rng(20261101)
    dev = torch.device("cuda")
    f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
    grid = engine.Grid(760.0, 1320.0, 28000)
    X = grid.axis()
    tau = 0.5 + 0.45 * np.sin(X / 11.0)
    La = 2.0 + np.cos(X / 23.0)
    Ld = 4.0 + 2.0 * np.sin(X / 9.0)
    cases = [dict(nk=60, nEnd=400, nMix=6, nPix=300, Q=4, hot=True),
             dict(nk=60, nEnd=7, nMix=3, nPix=333, Q=3, hot=True),
             dict(nk=2500, nEnd=5, nMix=2, nPix=65, Q=5, hot=False),
             dict(nk=60, nEnd=9, nMix=5, nPix=257, Q=6, hot=False),
             dict(nk=60, nEnd=31, nMix=4, nPix=130, Q=4, hot=False)]  # 39.7 KB of tables: the largest LDS case
    for cs in cases:
        Xk = np.sort(rng.uniform(750.0, 1330.0, cs["nk"]))
        em = rng.uniform(0.05, 1.0, (cs["nk"], cs["nEnd"]))
        kidx = rng.integers(0, cs["nEnd"], (cs["nPix"], cs["nMix"])).astype(np.int32)
        frac = rng.uniform(0.0, 1.0, (cs["nPix"], cs["nMix"]))
        frac /= frac.sum(1, keepdims=True)
        Tp = rng.uniform(270.0, 330.0, cs["nPix"])
        if cs["hot"]:
            Tp[rng.integers(0, cs["nPix"], 12)] = rng.uniform(1200.0, 1500.0, 12)
            Tp[-1] = 1450.0  # the last, partial workgroup takes the per-pixel path too
        xo, cube = sensor.hsi_cube(grid, f32(tau), f32(La), f32(Ld), Xk, f32(em), torch.as_tensor(kidx, device=dev), f32(frac),
                                   torch.as_tensor(Tp, device=dev), resFactor=2, Q=cs["Q"])
        em_hi = np.stack([np.interp(X, Xk, em[:, k]) for k in range(cs["nEnd"])], axis=1)
        em_p = np.einsum("pm,xpm->xp", frac, em_hi[:, kidx])
        L = tau[:, None] * (em_p * ref.planckian(X, Tp) + (1 - em_p) * Ld[:, None]) + La[:, None]
        xr, Lr = ref.ILS_MAKO(X, L, resFactor=2)
        assert np.array_equal(xo, xr) and cube.shape == Lr.shape, cs
        assert rel_err(cube.cpu().numpy(), Lr) <= TOL_L, cs


def test_fused_band_radiance_gaussian_and_cold_vs_oracle(rt):
    """sensor.band_radiance_fused (rtx_band_moments + rtx_band_mix) on the two paths the triangle / LWIR cases do not take:
    the Gaussian band shapes (ILS_MAKO.py:2-35; every point evaluated in fp64) and a surface so hot that a MAKO band's Planck
    exponent drops below 1.5 (planck_f32's expm1 branch instead of the per-task linear exponent)."""
    import torch
    from radtxfr_amd import engine, sensor
    rng = np.random.default_rng(20261102)
    dev = torch.device("cuda")
    f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
    grid = engine.Grid(745.0, 1335.0, 30000)
    X = grid.axis()
    tau = 0.5 + 0.45 * np.sin(X / 17.0)
    La = 2.0 + np.cos(X / 31.0)
    Ld = 4.0 + 2.0 * np.sin(X / 5.0)
    Xk = np.sort(rng.uniform(740.0, 1340.0, 90))
    em = rng.uniform(0.05, 1.0, (90, 4))
    em_hi = np.stack([np.interp(X, Xk, em[:, k]) for k in range(4)], axis=1)
    for Ts, kind in ((295.0, 1), (1400.0, 0), (1400.0, 1)):
        L_ref = ref.compute_LWIR_apparent_radiance(X, em_hi, np.array([Ts]), tau[:, None], La[:, None], Ld[:, None])[:, :, 0]
        xr, Lb_ref = ref.ILS_MAKO_gauss(X, L_ref) if kind else ref.ILS_MAKO(X, L_ref)
        xf, Lf = sensor.band_radiance_fused(grid, f32(tau), f32(La), f32(Ld), Xk, f32(em), Ts, kind=kind)
        assert np.array_equal(xf, xr), (Ts, kind)
        assert rel_err(Lf.cpu().numpy(), Lb_ref) <= TOL_L, (Ts, kind)
