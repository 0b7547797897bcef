"""Generate tests/golden/*.npz by RUNNING THE IMPORTED REFERENCE (build container only).

    python oracle/make_golden.py

The reference has no tests or golden vectors, so these files are the parity pin:
inputs + the reference's own outputs, small enough to commit. The reference never travels;
the fixtures and this script do. Fixture ids follow SURVEY.md 8c (G1..G7).
"""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
from _refimport import inject_table, load, make_oracle_OD  # noqa: E402
from radtxfr_amd import synthetic  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrs):
    p = os.path.join(OUT, name)
    np.savez_compressed(p, **arrs)
    print("%-28s %8.1f KB" % (name, os.path.getsize(p) / 1024))


def make_g9(rt):
    # ---- G9 smooth / reduceResolution (np.int restored for the call, NumPy >= 1.24 removed it) ----
    rng = np.random.default_rng(20261009)
    Xf = np.linspace(900.0, 904.0, 8001)  # 0.0005 cm^-1
    base = 1.0 + 0.5 * np.sin(7.0 * Xf) + 0.2 * np.cos(91.0 * Xf)
    for c0, wd in ((900.7, 0.004), (901.9, 0.02), (902.05, 0.0012), (903.3, 0.05)):
        base = base + 3.0 * wd * wd / ((Xf - c0) ** 2 + wd * wd)
    Y1 = base + 0.01 * rng.standard_normal(Xf.size)
    Y2 = np.stack([Y1, np.exp(-Y1), Y1[::-1] ** 2], axis=1)
    had = hasattr(np, "int")
    if not had:
        np.int = int
    try:
        sm11 = rt.smooth(Y1, 11, "hanning")
        sm50 = rt.smooth(Y1, 50, "hamming")
        smflat = rt.smooth(Y1, 7, "flat")
        Xo, Yo1 = rt.reduceResolution(Xf, Y1, 0.05)
        Yo2 = rt.reduceResolution(Xf, Y2, 0.05, X_out=Xo)
        Xo8, Yo8 = rt.reduceResolution(Xf, Y1, 0.02, N=8, window="blackman")
    finally:
        if not had:
            del np.int
    save("g9_reduce.npz", Xf=Xf, Y1=Y1, Y2=Y2, sm11=sm11, sm50=sm50, smflat=smflat, Xo=Xo, Yo1=Yo1, Yo2=Yo2, Xo8=Xo8, Yo8=Yo8)


def make_g10(hapi):
    """G10: absorptionCoefficient_Lorentz / _Doppler on the G4 table."""
    atm = synthetic.load_standard_atmosphere()
    tbl = synthetic.synth_line_table(synthetic.SEED_C2, 2000, 675.0, 1425.0)
    inject_table(hapi, "g10", tbl)
    out = {}
    gl = np.linspace(900.0, 960.0, 30001)  # 0.002 cm^-1
    for tag, row in (("l01", 0), ("l32", 31)):
        Tk, pk = float(atm[row, 5]), float(atm[row, 4]) / 101325.0
        _, xs_ = quiet(hapi.absorptionCoefficient_Lorentz, SourceTables="g10", Environment={"T": Tk, "p": pk}, OmegaGrid=gl)
        out["T_" + tag], out["p_" + tag], out["lor_" + tag] = Tk, pk, xs_
    _, out["lor_opt"] = quiet(hapi.absorptionCoefficient_Lorentz, Components=[(1, 1), (2, 1, 0.5)], SourceTables="g10",
                              Environment={"T": 250.0, "p": 0.4}, OmegaGrid=gl[5000:12000], HITRAN_units=False,
                              OmegaWing=1.0, OmegaWingHW=20.0, Diluent={"air": 0.7, "self": 0.3})
    gd = np.linspace(1000.0, 1003.0, 30001)  # 0.0001 cm^-1: Doppler HWHM here is ~1e-3
    for tag, (Tk, pk) in (("a", (296.0, 1.0)), ("b", (220.0, 0.05))):
        _, out["dop_" + tag] = quiet(hapi.absorptionCoefficient_Doppler, SourceTables="g10", Environment={"T": Tk, "p": pk},
                                     OmegaGrid=gd)
    _, out["dop_noshift"] = quiet(hapi.absorptionCoefficient_Doppler, SourceTables="g10", Environment={"T": 296.0, "p": 1.0},
                                  OmegaGrid=gd, LineShift=False, HITRAN_units=False, OmegaWing=0.05)
    save("g10_lorentz_doppler.npz", seed=synthetic.SEED_C2, n_lines=2000, nu_lo=675.0, nu_hi=1425.0, gl_lo=900.0, gl_hi=960.0,
         gl_n=30001, gd_lo=1000.0, gd_hi=1003.0, gd_n=30001, **out)


def make_g11(hapi):
    """G11: absorptionCoefficient_SDVoigt with speed-dependence columns."""
    tbl = dict(synthetic.synth_line_table(synthetic.SEED_C2, 400, 895.0, 917.0))
    rng = np.random.default_rng(20261017)
    tbl["SD_air"] = rng.uniform(0.05, 0.2, 400)
    tbl["SD_self"] = rng.uniform(0.0, 0.1, 400)
    tbl["SD_air"][::7] = 0.0  # some lines without speed dependence (PART1 for those)
    tbl["SD_self"][::7] = 0.0
    inject_table(hapi, "g11", tbl)
    g = np.linspace(900.0, 912.0, 6001)
    out = {}
    for tag, kw in (("a", dict(Environment={"T": 250.0, "p": 0.3})),
                    ("b", dict(Environment={"T": 296.0, "p": 1.0}, Diluent={"air": 0.6, "self": 0.4})),
                    ("c", dict(Environment={"T": 220.0, "p": 0.01})),
                    ("d", dict(Environment={"T": 300.0, "p": 0.8}, HITRAN_units=False, OmegaWing=0.5, OmegaWingHW=20.0,
                               Components=[(1, 1), (2, 1, 0.5)]))):
        _, out["xs_" + tag] = quiet(hapi.absorptionCoefficient_SDVoigt, SourceTables="g11", OmegaGrid=g, **kw)
    save("g11_sdvoigt.npz", seed=synthetic.SEED_C2, n_lines=400, nu_lo=895.0, nu_hi=917.0, g_lo=900.0, g_hi=912.0, g_n=6001,
         SD_air=tbl["SD_air"], SD_self=tbl["SD_self"], **out)


def make_g12(hapi):
    """G12: the reference's own .par/.header storage layer (db_begin -> loadCache -> storage2cache ->
    getRowObjectFromString, misc/hapi.py:1535-1672, 1718-1730, 5205-5221) on two synthetic tables whose text is
    committed next to the fixture (tests/golden/g12_*.par|.data|.header), and its Voigt line-sum on the parsed rows.
      g12a.par                no header (the reference writes the default 160-character one); rows with isotopologue
                              codes '0', 'A', 'B', dropped leading zeros, blank g'/g'' fields, a short and a blank line
      g12b.data + .header     default fields + comma-separated extras n_self, deltap_air, delta_self, deltap_self
    """
    import json
    import shutil
    import tempfile

    from radtxfr_amd import hitran_par

    rng = np.random.default_rng(20261012)
    tbl = synthetic.synth_line_table(20261012, 60, 990.0, 1010.0)
    n = tbl["nu"].size
    tbl = dict(tbl)
    tbl["local_iso_id"] = np.where(np.arange(n) % 11 == 5, 0, tbl["local_iso_id"])      # '0': the tenth isotopologue
    tbl["molec_id"] = np.where(tbl["local_iso_id"] == 0, 2, tbl["molec_id"])              # (2,0) exists in ISO / TIPS
    tbl["delta_air"] = np.where(np.arange(n) % 7 == 3, -0.0123, tbl["delta_air"])        # '-.012300'
    tbl["gamma_air"] = np.where(np.arange(n) % 9 == 2, 0.1234, tbl["gamma_air"])         # '.1234'
    tbl["a"] = 10.0 ** rng.uniform(-3, 2, n)
    tbl["gp"] = rng.integers(1, 60, n).astype(float)
    tbl["gpp"] = rng.integers(1, 60, n).astype(float)
    tbl["global_upper_quanta"] = ["      0 1 0    "] * n
    tbl["local_lower_quanta"] = ["  %2d  %1d  %1d     " % (k % 30, k % 5, k % 3) for k in range(n)]
    tmp = tempfile.mkdtemp(prefix="g12_")
    try:
        pa = os.path.join(tmp, "g12a.par")
        hitran_par.write_par(pa, tbl)
        rows = open(pa).read().split("\n")[:-1]
        assert all(len(r) == 160 for r in rows)
        edit = list(rows)
        edit[4] = edit[4][:2] + "A" + edit[4][3:]          # letter isotopologue code: int('A') fails -> row dropped
        edit[9] = edit[9][:2] + "B" + edit[9][3:]
        edit[13] = edit[13][:146] + " " * 14                # blank g' / g'': float('       ') fails -> row dropped
        edit[21] = edit[21][:100]                           # short record
        edit.insert(30, "")                                 # blank line
        edit[40] = edit[40][:15] + " 1.234e-21" + edit[40][25:]   # lower-case exponent
        text_a = "\n".join(edit) + "\n"
        open(pa, "w").write(text_a)
        # table b: explicit header with comma-separated extras
        hdr = json.loads(json.dumps(hapi.HITRAN_DEFAULT_HEADER))
        hdr["table_name"] = "g12b"
        hdr["extra"] = ["n_self", "deltap_air", "delta_self", "deltap_self"]
        hdr["extra_format"] = {"n_self": "%7.4f", "deltap_air": "%10.3E", "delta_self": "%9.6f", "deltap_self": "%10.3E"}
        hdr["extra_separator"] = ","
        ex = {"n_self": np.round(rng.uniform(0.0, 0.9, n) * (np.arange(n) % 4 != 1), 4),
              "deltap_air": rng.uniform(-2e-5, 2e-5, n), "delta_self": np.round(rng.uniform(-0.02, 0.01, n), 6),
              "deltap_self": rng.uniform(-5e-5, 5e-5, n)}
        text_b = ""
        for k, r in enumerate(rows):
            tail = ",%7.4f,%10.3E,%9.6f,%10.3E" % (ex["n_self"][k], ex["deltap_air"][k], ex["delta_self"][k], ex["deltap_self"][k])
            if k == 17:
                tail = ",   oops,%10.3E,%9.6f,%10.3E" % (ex["deltap_air"][k], ex["delta_self"][k], ex["deltap_self"][k])  # -> 0.0
            if k == 33:
                tail = ",0.5"                                                                  # too few chunks: dropped
            text_b += r + tail + "\n"
        open(os.path.join(tmp, "g12b.data"), "w").write(text_b)
        open(os.path.join(tmp, "g12b.header"), "w").write(json.dumps(hdr, indent=2))
        quiet(hapi.db_begin, tmp)
        out = {}
        for name in ("g12a", "g12b"):
            T = hapi.LOCAL_TABLE_CACHE[name]
            out[name + "_nrows"] = T["header"]["number_of_rows"]
            out[name + "_order"] = np.array(T["header"]["order"])
            for col, v in T["data"].items():
                out[name + "_" + col] = np.array(v)
        # the reference's line-sum on what it parsed (air + self diluents: exercises n_self / deltap_* / delta_self)
        grid = np.linspace(995.0, 1005.0, 5001)
        for name, dil in (("g12a", {"air": 1.0}), ("g12b", {"air": 0.6, "self": 0.4})):
            _, xs = quiet(hapi.absorptionCoefficient_Voigt, SourceTables=name, Environment={"T": 251.3, "p": 0.7},
                          OmegaGrid=grid, HITRAN_units=True, Diluent=dil)
            out[name + "_xs"] = xs
        save("g12_par_tables.npz", grid_lo=995.0, grid_hi=1005.0, grid_n=5001, T=251.3, p=0.7, **out)
        # the input text travels as data files next to the fixture
        shutil.copy(pa, os.path.join(OUT, "g12a.par"))
        shutil.copy(os.path.join(tmp, "g12b.data"), os.path.join(OUT, "g12b.data"))
        shutil.copy(os.path.join(tmp, "g12b.header"), os.path.join(OUT, "g12b.header"))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def make_g13(rt, hapi):
    """G13: compute_TUD with a VECTOR theta_r (radiative_transfer.py:313, 346-365): 2 sensor altitudes x 2 slant paths
    -> tau, L-up of shape (nX, 2, 2); 1 altitude x 3 slant paths -> (nX, 3). Thin column (as G8) so tau spans (0,1)."""
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    xlo, xhi = 1000.0, 1000.8
    sub = synthetic.subset_table(full, xlo - 12.0, xhi + 12.0)
    inject_table(hapi, "g13", sub)
    rt.compute_OD = make_oracle_OD(hapi, "g13", sub)
    a = synthetic.c3_atmosphere(32)
    opts = dict(DVOUT=0.001, Zs=a["Zs"], Ts=a["Ts"], Ps=a["Ps"], PLs=a["PLs"], MFs_VAL=a["MFs_VAL"] * 1e-3, MFs_ID=a["MFs_ID"],
                N_angle=30, save=False, returnOD=False)
    th22, alt22 = np.array([0.0, 0.7]), np.array([2.0, 9.0])
    _, tau22, Lu22, Ld22 = quiet(rt.compute_TUD, xlo, xhi, **dict(opts, theta_r=th22, Altitudes=alt22))
    th13 = np.array([0.2, 0.5, 1.0])
    _, tau13, Lu13, Ld13 = quiet(rt.compute_TUD, xlo, xhi, **dict(opts, theta_r=th13, Altitudes=np.asarray([500])))
    assert tau22.shape[1:] == (2, 2) and tau13.shape[1:] == (3,)
    save("g13_tud_slants.npz", seed=synthetic.SEED_C3, n_lines=100000, nu_lo=475.0, nu_hi=6025.0, pad=12.0, lo=xlo, hi=xhi,
         mf_scale=1e-3, th22=th22, alt22=alt22, tau22=tau22, Lu22=Lu22, Ld22=Ld22, th13=th13, tau13=tau13, Lu13=Lu13, Ld13=Ld13)


def main():
    os.makedirs(OUT, exist_ok=True)
    rt, hapi, ils_gauss = load()
    atm = synthetic.load_standard_atmosphere()

    # ---- G0 module defaults (embedded StdAtmos table and the options built from it) ---------
    save("g0_defaults.npz", StdAtmos=rt.StdAtmos, Zs=rt.options["Zs"], Ts=rt.options["Ts"], Ps=rt.options["Ps"],
         PLs=rt.options["PLs"], MFs_VAL=rt.options["MFs_VAL"], MFs_ID=rt.options["MFs_ID"],
         DVOUT=rt.options["DVOUT"], N_angle=rt.options["N_angle"], Altitudes=rt.options["Altitudes"],
         c1=rt.c1, c2=rt.c2)

    # ---- G1 planckian -----------------------------------------------------------------
    X = np.linspace(500, 6000, 56)
    T32 = atm[:32, 5]
    Xum = np.linspace(7.5, 13.5, 25)
    T2d = np.array([[250.0, 275.0, 300.0], [216.65, 288.15, 320.0]])
    save("g1_planck.npz", X=X, T32=T32, L32=rt.planckian(X, T32),
         Ls=rt.planckian(X, 296.0), Xum=Xum, Lum=rt.planckian(Xum, T32[:5], wavelength=True),
         T2d=T2d, L2d=rt.planckian(X, T2d),
         spot=np.array([rt.planckian(500, 296)[0], *rt.planckian(1000, [250, 300])[0],
                        rt.planckian(10.0, 300, wavelength=True)[0]]),
         BT=rt.brightnessTemperature(X, rt.planckian(X, T32)),
         BTum=rt.brightnessTemperature(Xum, rt.planckian(Xum, T32[:5], wavelength=True), wavelength=True),
         L_bt2l=rt.BT2L(X, np.tile(T32[None, :8], (X.size, 1))))

    # ---- G2 complex probability function / Voigt profile ---------------------------------
    xs = np.concatenate([np.linspace(-20, 20, 81), [14.9, 14.99, 15.0, 15.01, -14.95]])
    ys = np.array([1e-4, 1e-2, 0.1, 0.5, 1.0, 3.0, 8.0, 14.0, 14.999, 15.0, 25.0, 60.0])
    xg, yg = np.meshgrid(xs, ys, indexing="ij")
    wr, wi = hapi.hum1_wei(xg.ravel().copy(), yg.ravel().copy())
    sg = np.array([999, 999.9, 1000, 1000.05, 1003.5])
    pv = hapi.PROFILE_VOIGT(1000, 0.0009, 0.07, sg)[0]
    sg2 = np.linspace(2349.0, 2351.0, 401)
    pv2 = hapi.PROFILE_VOIGT(2350.0123, 0.0022, 0.004, sg2)[0]
    # Weideman-24 coefficients exactly as the reference rebuilds them per call (:9816-9824)
    N = 24
    M_ = 2 * N
    k = np.arange(-M_ + 1, M_)
    L = np.sqrt(N / np.sqrt(2))
    t = L * np.tan(k * np.pi / M_ / 2)
    f = np.zeros(len(t) + 1)
    f[1:] = np.exp(-t ** 2) * (L ** 2 + t ** 2)
    a = np.flipud(np.real(np.fft.fft(np.fft.fftshift(f)))[1:N + 1] / (2 * M_))
    save("g2_cpf_voigt.npz", x=xg.ravel(), y=yg.ravel(), wr=wr, wi=wi, sg=sg, pv=pv, sg2=sg2, pv2=pv2,
         w24=a, L24=np.array(L))

    # ---- G3 TIPS ------------------------------------------------------------------------
    mis = [(1, 1), (1, 2), (2, 1), (2, 2), (3, 1)]
    Ts = np.concatenate([T32, [296.0, 70.0, 84.9, 85.0, 2990.0, 3000.0]])
    Q = np.array([[hapi.PYTIPS(m, i, float(tt)) for tt in Ts] for (m, i) in mis])
    Sx = hapi.EnvironmentDependency_Intensity(1e-21, 250., 296., hapi.PYTIPS(1, 1, 250.), hapi.PYTIPS(1, 1, 296.), 500., 1000.)
    save("g3_tips.npz", mi=np.array(mis), T=Ts, Q=Q, S_spot=np.array(Sx))

    # ---- G4 absorptionCoefficient_Voigt, 2000-line table, C2 grid --------------------------
    tbl = synthetic.synth_line_table(synthetic.SEED_C2, 2000, 675.0, 1425.0)
    inject_table(hapi, "g4", tbl)
    grid = np.linspace(700, 1400, 70000)
    out = {}
    for tag, row in (("l01", 0), ("l32", 31)):
        Tk, pk = float(atm[row, 5]), float(atm[row, 4]) / 101325.0
        _, xs_ = quiet(hapi.absorptionCoefficient_Voigt, SourceTables="g4", Environment={"T": Tk, "p": pk},
                       OmegaGrid=grid, HITRAN_units=True)
        out["T_" + tag], out["p_" + tag], out["xs_" + tag] = Tk, pk, xs_
    # options exercised: explicit Components with custom abundance, self-broadening, HITRAN_units=False, wings
    _, xs_opt = quiet(hapi.absorptionCoefficient_Voigt, Components=[(1, 1), (2, 1, 0.5)], SourceTables="g4",
                      Environment={"T": 250.0, "p": 0.4}, OmegaGrid=grid[20000:30000], HITRAN_units=False,
                      GammaL="gamma_self", OmegaWing=2.0, OmegaWingHW=20.0)
    _, xs_dil = quiet(hapi.absorptionCoefficient_Voigt, SourceTables="g4", Environment={"T": 230.0, "p": 0.05},
                      OmegaGrid=grid[40000:46000], Diluent={"air": 0.7, "self": 0.3})
    save("g4_voigt_xsec.npz", seed=synthetic.SEED_C2, n_lines=2000, nu_lo=675.0, nu_hi=1425.0,
         grid_lo=700.0, grid_hi=1400.0, grid_n=70000, xs_opt=xs_opt, xs_dil=xs_dil,
         **{k_: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k_, v in out.items()})

    # ---- G5 per-layer OD + compute_TUD on 3 windows x 32 layers ---------------------------
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    a32 = synthetic.c3_atmosphere(32)
    g5 = {}
    for w, (xlo, xhi) in enumerate([(500.0, 502.0), (2349.0, 2351.0), (5998.0, 6000.0)]):
        sub = synthetic.subset_table(full, xlo - 12.0, xhi + 12.0)
        name = "g5w%d" % w
        inject_table(hapi, name, sub)
        rt.compute_OD = make_oracle_OD(hapi, name, sub)
        opts = dict(DVOUT=0.001, Zs=a32["Zs"], Ts=a32["Ts"], Ps=a32["Ps"], PLs=a32["PLs"], MFs_VAL=a32["MFs_VAL"],
                    MFs_ID=a32["MFs_ID"], theta_r=0, N_angle=30, Altitudes=np.asarray([500]), save=True, returnOD=False)
        cwd = os.getcwd()
        os.chdir("/tmp")
        try:
            Xw, tau, Lu, Ld = quiet(rt.compute_TUD, xlo, xhi, **opts)
            dump = np.load("/tmp/ComputeTUD.npz")
            OD = dump["OD"]
        finally:
            os.chdir(cwd)
        g5.update({f"w{w}_lo": xlo, f"w{w}_hi": xhi, f"w{w}_X": Xw, f"w{w}_tau": tau, f"w{w}_Lu": Lu,
                   f"w{w}_Ld": Ld, f"w{w}_OD": OD})
        if w == 1:
            # quirk coverage: slant path, several sensor altitudes, returnOD, fewer angles
            o2 = dict(opts, theta_r=0.6, Altitudes=np.asarray([1.0, 4.05, 9.0]), N_angle=7, save=False)
            _, tau2, Lu2, Ld2 = quiet(rt.compute_TUD, xlo, xhi, **o2)
            o3 = dict(opts, Altitudes=np.asarray([9.0]), returnOD=True, save=False)
            _, tau3, Lu3, Ld3 = quiet(rt.compute_TUD, xlo, xhi, **o3)
            g5.update(w1_tau_alt=tau2, w1_Lu_alt=Lu2, w1_Ld_alt=Ld2, w1_tau_rod=tau3, w1_Lu_rod=Lu3, w1_Ld_rod=Ld3)
    save("g5_tud_windows.npz", seed=synthetic.SEED_C3, n_lines=100000, nu_lo=475.0, nu_hi=6025.0, pad=12.0, **g5)

    # ---- G8 optically THIN atmosphere (the SURVEY 8d table is nearly opaque everywhere: tau ~ 0). Mixing
    # ratios scaled by 1e-3 put layer optical depths in 1e-3..10, so tau spans (0,1) and L-up/L-down weigh
    # many layers; all 66 layers of the standard atmosphere (Doppler-dominated upper layers, y << 1).
    g8 = {}
    for tag, (xlo, xhi, nlay, scale) in {"a": (1000.0, 1002.0, 32, 1e-3), "b": (2380.0, 2381.0, 66, 3e-4)}.items():
        sub = synthetic.subset_table(full, xlo - 12.0, xhi + 12.0)
        name = "g8" + tag
        inject_table(hapi, name, sub)
        rt.compute_OD = make_oracle_OD(hapi, name, sub)
        A = atm[:nlay]
        opts = dict(DVOUT=0.001, Zs=A[:, 1], Ts=A[:, 5], Ps=A[:, 4], PLs=A[:, 3], MFs_VAL=A[:, 6:8] * 1e6 * scale,
                    MFs_ID=np.array([1, 2]), theta_r=0.3, N_angle=30, Altitudes=np.asarray([500]), save=False, returnOD=False)
        Xw, tau, Lu, Ld = quiet(rt.compute_TUD, xlo, xhi, **opts)
        g8.update({f"{tag}_lo": xlo, f"{tag}_hi": xhi, f"{tag}_nlay": nlay, f"{tag}_scale": scale, f"{tag}_tau": tau,
                   f"{tag}_Lu": Lu, f"{tag}_Ld": Ld})
    save("g8_tud_thin.npz", seed=synthetic.SEED_C3, n_lines=100000, nu_lo=475.0, nu_hi=6025.0, pad=12.0, theta_r=0.3, **g8)

    # ---- G6 apparent radiance --------------------------------------------------------------
    rng = np.random.default_rng(7)
    nX, nE, nA = 128, 9, 3
    Xb = np.sort(1e4 / np.linspace(7.6, 13.1, nX))
    emis = rng.uniform(0.6, 0.99, (nX, nE))
    Tsurf = np.array([280.0, 287.87, 301.5])
    tau = rng.uniform(0.2, 0.98, (nX, nA))
    La = rng.uniform(0.5, 4.0, (nX, nA))
    Ld = rng.uniform(1.0, 8.0, (nX, nA))
    dT = np.arange(-10, 10.5, 2.5)
    L0 = rt.compute_LWIR_apparent_radiance(Xb, emis, Tsurf, tau, La, Ld)
    L1, Ls1 = rt.compute_LWIR_apparent_radiance(Xb, emis, Tsurf, tau, La, Ld, dT=dT, return_Ls=True)
    save("g6_apparent_radiance.npz", X=Xb, emis=emis, Ts=Tsurf, tau=tau, La=La, Ld=Ld, dT=dT, L0=L0, L1=L1, Ls1=Ls1)

    # ---- G7 ILS ---------------------------------------------------------------------------
    Xh = np.linspace(740.0, 1340.0, 24000)
    Y1 = 5 + np.sin(Xh / 7.0) + 0.3 * np.cos(Xh * 3.1)
    Y2 = np.stack([Y1, np.exp(-((Xh - 1000) / 80.0) ** 2), rng.uniform(0, 1, Xh.size)], axis=1)
    xo1, yo1 = rt.ILS_MAKO(Xh, Y1)
    xo2, yo2 = rt.ILS_MAKO(Xh, Y2)
    xo3, yo3 = rt.ILS_MAKO(Xh, Y2, resFactor=2)
    yo4 = rt.ILS_MAKO(Xh, Y2, returnX=False, fwhm_sf=1.3, shift=0.4, scale=1.0005)
    xg1, yg1 = ils_gauss.ILS_MAKO(Xh, Y1)
    xg2, yg2 = ils_gauss.ILS_MAKO(Xh, Y2)
    save("g7_ils.npz", X_lo=740.0, X_hi=1340.0, X_n=24000, Y2=Y2.astype(np.float64), xo1=xo1, yo1=yo1, xo2=xo2,
         yo2=yo2, xo3=xo3, yo3=yo3, yo4=yo4, xg1=xg1, yg1=yg1, xg2=xg2, yg2=yg2)

    make_g9(rt)
    make_g10(hapi)
    make_g11(hapi)
    make_g12(hapi)
    make_g13(rt, hapi)


if __name__ == "__main__":
    if sys.argv[1:] == ["g9"]:  # regenerate only one of the newer fixtures
        os.makedirs(OUT, exist_ok=True)
        make_g9(load()[0])
    elif sys.argv[1:] == ["g10"]:
        os.makedirs(OUT, exist_ok=True)
        make_g10(load()[1])
    elif sys.argv[1:] == ["g11"]:
        os.makedirs(OUT, exist_ok=True)
        make_g11(load()[1])
    elif sys.argv[1:] == ["g12"]:
        os.makedirs(OUT, exist_ok=True)
        make_g12(load()[1])
    elif sys.argv[1:] == ["g13"]:
        os.makedirs(OUT, exist_ok=True)
        r_ = load()
        make_g13(r_[0], r_[1])
    else:
        main()
