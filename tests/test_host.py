"""CPU-only checks of the host logic and of the C-ABI boundary (no compute calls: no GPU here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib_or_skip():
    from radtxfr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib


def test_library_loads_and_exports_every_declared_symbol():
    _lib = _lib_or_skip()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "radtxfr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rtx_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rtx_version() == 100
    assert lib.rtx_voigt_tile_points() % 256 == 0


def test_argument_errors_come_back_as_text_not_crashes():
    import ctypes as C
    _lib = _lib_or_skip()
    lib = _lib.load()
    g = _lib.make_grid(500.0, 501.0, 1001)
    # bad grid / NULL pointers are rejected before any device work
    bad = _lib.make_grid(500.0, 501.0, 1001, offset=900, n=500)
    assert lib.rtx_voigt_sum(None, C.byref(g), 1, None, None, 1001, None) != 0
    assert b"NULL" in lib.rtx_last_error()
    assert lib.rtx_planck(C.byref(bad), None, 500, None, 1, 0, None, None) != 0
    assert b"outside" in lib.rtx_last_error()
    with pytest.raises(_lib.RtxError):
        _lib.check(lib.rtx_ils(7, C.byref(g), None, 1001, None, 1, 1, 1, None, None, None, None))


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from radtxfr_amd import _lib
    from radtxfr_amd import radiative_transfer as rt
    with pytest.raises(_lib.RtxError):
        rt.planckian(np.linspace(800, 1200, 5), 300.0)
    with pytest.raises(_lib.RtxError):
        rt.ILS_MAKO(np.linspace(760, 1320, 2000), np.ones(2000))


def test_product_never_imports_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "radtxfr_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "cpu_ref" not in src, f


def test_make_grid_matches_linspace():
    from radtxfr_amd import _lib
    for a, b, n in ((500.0, 6000.0, 5500000), (700.0, 1400.0, 70000), (2349.0, 2351.0, 2000), (0.1, 0.7, 7)):
        X = np.linspace(a, b, n)
        g = _lib.make_grid(a, b, n)
        idx = np.unique(np.concatenate([np.arange(0, min(n, 50)), np.arange(max(0, n - 50), n - 1),
                                        np.random.default_rng(0).integers(0, n - 1, 1000)]))
        assert np.array_equal(idx * g.step + g.xmin, X[idx])  # arange*step+start, two roundings
        assert X[-1] == g.xmax


def test_grid_from_axis():
    from radtxfr_amd.engine import Grid
    X = np.linspace(700, 1400, 70000)
    g = Grid.from_axis(X)
    assert (g.n_total, g.offset, g.n) == (70000, 0, 70000)
    assert np.array_equal(g.shard(10, 100).axis(), X[10:110])
    with pytest.raises(NotImplementedError):
        Grid.from_axis(np.sort(1e4 / np.linspace(7.5, 13.5, 100)))


def test_tips_against_golden(golden):
    from radtxfr_amd import tips
    g = golden("g3_tips.npz")
    for r, (m, i) in enumerate(g["mi"].tolist()):
        q = [tips.PYTIPS(m, i, float(t)) for t in g["T"]]
        np.testing.assert_allclose(q, g["Q"][r], rtol=1e-14)
    with pytest.raises(Exception, match="between 70K and 3000K"):
        tips.PYTIPS(1, 1, 3000.5)
    with pytest.raises(Exception, match="no data"):
        tips.PYTIPS(77, 1, 296.0)
    assert tips.abundance(1, 1) == 0.997317 and tips.molecularMass(2, 1) == 43.98983


def test_weideman_include_matches_golden(golden):
    g = golden("g2_cpf_voigt.npz")
    src = open(os.path.join(ROOT, "radtxfr_amd", "csrc", "w24_coeffs.inc")).read()
    body = src[src.index("W24D[24] = {") + 12: src.index("};")]
    vals = np.array([float(v) for v in body.replace("\n", " ").split(",")])
    assert np.array_equal(vals, g["w24"])
    L = float(re.search(r"#define W24_L (\S+)", src).group(1))
    assert L == float(g["L24"])


def test_module_defaults_match_reference(golden):
    from radtxfr_amd import radiative_transfer as rt
    g = golden("g0_defaults.npz")
    assert np.array_equal(rt.StdAtmos, g["StdAtmos"])
    for k in ("Zs", "Ts", "Ps", "PLs", "MFs_VAL", "MFs_ID", "Altitudes"):
        assert np.array_equal(np.asarray(rt.options[k]), g[k]), k
    assert rt.options["DVOUT"] == float(g["DVOUT"]) and rt.options["N_angle"] == int(g["N_angle"])
    assert rt.c1 == float(g["c1"]) and rt.c2 == float(g["c2"])
    X = rt.make_spectral_axis(500.0, 6000.0, 0.001)
    assert X.size == 5500000 and X[0] == 500.0 and X[-1] == 6000.0


def test_hapi_host_helpers():
    from radtxfr_amd import hapi
    om = hapi.arange_(995.0, 1005.0, 0.01)
    assert om.size == 1001 and om[0] == 995.0
    assert hapi.listOfTuples((1, 1)) == (1, 1) and hapi.listOfTuples("a") == ["a"]
    assert abs(hapi.volumeConcentration(1.0, 296.0) - 2.4794e19) / 2.4794e19 < 1e-3


def test_layer_weights_follow_the_od_contract():
    """weight = n(p,T) * x_m * PL*1e5 for the line's molecule; molecules not in MF_ID contribute nothing."""
    from radtxfr_amd import engine
    sp = [(1, 1), (1, 2), (2, 1), (6, 1)]
    T, P, PL = np.array([288.0, 250.0]), np.array([101325.0, 50000.0]), np.array([0.1, 0.5])
    MF = np.array([[7000.0, 380.0], [3000.0, 381.0]])
    w, p_atm = engine.layer_weights_od(sp, T, P, PL, MF, [1, 2])
    n = engine.volumeConcentration(P / 101325.0, T)
    np.testing.assert_allclose(w[0], n * MF[:, 0] * 1e-6 * PL * 1e5, rtol=1e-15)
    np.testing.assert_array_equal(w[0], w[1])
    np.testing.assert_allclose(w[2], n * MF[:, 1] * 1e-6 * PL * 1e5, rtol=1e-15)
    assert np.all(w[3] == 0) and np.allclose(p_atm, [1.0, 50000.0 / 101325.0])


def _check_table_against_golden(cols, header, g, name):
    order = [str(x) for x in g[name + "_order"]]
    assert header["order"] == order and header["number_of_rows"] == int(g[name + "_nrows"])
    for k in order:
        want = g[name + "_" + k]
        if want.dtype.kind in "US":
            assert list(cols[k]) == [str(x) for x in want], k
        else:
            got = np.asarray(cols[k])
            assert got.dtype.kind == want.dtype.kind and np.array_equal(got, want), k  # bit-exact: same int()/float()


def test_par_reader_equals_reference_storage2cache(golden):
    """SURVEY 8f row 3. tests/golden/g12a.par and g12b.data/.header are synthetic line files; g12_par_tables.npz holds
    what the REFERENCE's db_begin -> storage2cache -> getRowObjectFromString (misc/hapi.py:1535-1672) made of them
    (oracle/make_golden.py:make_g12). Every column must agree exactly, including the rows the reference drops:
    isotopologue codes 'A'/'B' (int() fails), blank g'/g'' fields, a truncated record, a blank line, a row with too
    few comma-separated extras -- and what it keeps: code '0' -> local_iso_id 0, '-.012300', '.1234', ' 1.234e-21',
    an unparsable extra -> 0.0."""
    from radtxfr_amd import hapi, hitran_par
    from conftest import GOLDEN
    g = golden("g12_par_tables.npz")
    ha, ca = hitran_par.read_table(os.path.join(GOLDEN, "g12a.par"))
    _check_table_against_golden(ca, ha, g, "g12a")
    assert 0 in ca["local_iso_id"] and ca["delta_air"].min() == -0.0123 and 0.1234 in ca["gamma_air"] and 1.234e-21 in ca["sw"]
    hb, cb = hitran_par.read_table(os.path.join(GOLDEN, "g12b.data"))
    _check_table_against_golden(cb, hb, g, "g12b")
    assert "extra" not in hb and hb["format"]["deltap_self"] == "%10.3E"
    # db_begin: the folder scan of loadCache (misc/hapi.py:1718-1730) -- headers first, then header-less .par files
    names = hapi.db_begin(GOLDEN)
    assert names == ["g12b", "g12a"] and set(names) <= set(hapi.tableList())
    assert hapi.LOCAL_TABLE_CACHE["g12a"]["header"]["number_of_rows"] == int(g["g12a_nrows"])
    assert np.array_equal(hapi.LOCAL_TABLE_CACHE["g12b"]["data"]["n_self"], g["g12b_n_self"])
    for n in names:
        hapi.LOCAL_TABLE_CACHE.pop(n)


def test_par_writer_records_parse_back(tmp_path):
    """write_par emits full 160-character records (g'/g'' as %7.1f: left blank, the reference's parser drops the row);
    synthetic tables are already rounded to .par precision, so the round trip through the vectorised reader is exact.
    A HITRAN-style record typed by hand parses field by field (columns per HITRAN_FORMAT_160, misc/hapi.py:468-489)."""
    from radtxfr_amd import hapi, hitran_par, synthetic
    tbl = synthetic.synth_line_table(7, 500, 600.0, 1400.0)
    tbl["local_iso_id"][:2] = [0, 9]
    p = tmp_path / "t.par"
    hitran_par.write_par(str(p), tbl)
    lines = open(p).read().splitlines()
    assert len(lines) == 500 and all(len(ln) == 160 for ln in lines)
    got = hitran_par.storage2cache("t", str(p))
    for k in ("molec_id", "local_iso_id", "nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air"):
        assert np.array_equal(got[k], tbl[k]), k
    assert hapi.LOCAL_TABLE_CACHE.pop("t")["header"]["number_of_rows"] == 500
    with pytest.raises(ValueError):
        hitran_par.write_par(str(p), dict(tbl, local_iso_id=np.full(500, 11)))
    rec = (" 2" + "1" + "  667.661431" + " 1.152E-19" + " 1.518E+00" + ".0720" + "0.094" + "    3.9032" + "0.75" + "-.000870"
           + " " * 79 + "    3.0" + "    5.0")
    assert len(rec) == 160
    q = tmp_path / "one.par"
    q.write_text(rec + "\n" + rec[:146] + "\n")  # second record without g'/g'': dropped
    one = hitran_par.read_par(str(q))
    assert one["nu"].size == 1 and (one["molec_id"][0], one["local_iso_id"][0]) == (2, 1)
    assert one["nu"][0] == 667.661431 and one["sw"][0] == 1.152e-19 and one["gamma_air"][0] == 0.072
    assert one["gamma_self"][0] == 0.094 and one["elower"][0] == 3.9032 and one["n_air"][0] == 0.75 and one["delta_air"][0] == -0.00087
    assert one["gp"][0] == 3.0 and one["gpp"][0] == 5.0


def test_species_factors_skip_unselected_species():
    """misc/hapi.py:11066: a line whose (M, I) is not among the Components is skipped before PYTIPS / molecularMass are
    called, so an isotopologue without TIPS data only raises when it is actually selected."""
    from radtxfr_amd import engine
    species = [(1, 1), (2, 1), (99, 1)]
    T = np.array([250.0, 296.0])
    w = np.array([[1.0, 1.0], [0.5, 0.0], [0.0, 0.0]])
    q, mass = engine.species_factors(species, T, weight=w)
    assert q[2].tolist() == [1.0, 1.0] and mass[2] == 1.0 and q[0, 1] == 1.0 and q[0, 0] > 1.0
    with pytest.raises(Exception):
        engine.species_factors(species, T)
    with pytest.raises(Exception):
        engine.species_factors(species, np.array([60.0]), weight=np.ones((3, 1)))


def test_vectorised_tips_is_bit_identical_to_the_scalar_routine():
    """tips.partition_sums (one NumPy pass per atmosphere, inside bench.py's timed step) == tips.partition_sum (the
    statement-by-statement restatement of AtoB, misc/hapi.py:5311-5388, pinned by golden G3) to the last bit, including
    the 3-point branches at both ends of the table and the 70 / 3000 K limits."""
    from radtxfr_amd import synthetic, tips
    rng = np.random.default_rng(5)
    species = [(1, 1), (1, 2), (2, 1), (2, 2), (3, 1), (2, 0)]
    T = np.concatenate([synthetic.c3_atmosphere(32)["Ts"], [70.0, 84.9, 85.0, 85.1, 109.99, 110.0, 296.0, 2985.0, 2990.0, 3000.0],
                        rng.uniform(70.0, 3000.0, 300)])
    want = np.array([[tips.partition_sum(m, i, t) for t in T] for m, i in species])
    assert np.array_equal(tips.partition_sums(species, T), want)
    mid = rng.uniform(120.0, 2900.0, 64)  # fast path: every temperature inside the table
    assert np.array_equal(tips.partition_sums(species, mid), np.array([[tips.partition_sum(m, i, t) for t in mid] for m, i in species]))
    plan = tips.PartitionPlan(species)  # fixed species list, constant Lagrange denominators: what engine.species_factors uses
    assert np.array_equal(plan(T), want) and np.array_equal(plan(mid), tips.partition_sums(species, mid))
    with pytest.raises(Exception):
        plan(np.array([250.0, 3000.1]))
    with pytest.raises(Exception):
        tips.partition_sums(species, np.array([250.0, 69.9]))
    with pytest.raises(Exception):
        tips.partition_sums([(99, 1)], np.array([250.0]))


def test_device_table_signature_covers_every_column():
    """hapi._device_table's cache key is a content fingerprint of all uploaded columns: editing sw (or any other
    column) in place, or swapping two rows, changes it (id(nu) + row count alone went stale)."""
    from radtxfr_amd import hapi
    a = np.linspace(1.0, 2.0, 1000)
    s0 = hapi._column_signature(a, 1000)
    b = a.copy()
    b[[3, 700]] = b[[700, 3]]
    assert hapi._column_signature(b, 1000) != s0 and hapi._column_signature(a * 1.0000001, 1000) != s0
    assert hapi._column_signature(a.tolist(), 1000) == s0 and hapi._column_signature(a, 999) != s0


def test_chebyshev_tables_match_generator_and_error_bounds():
    """cheb8_64.inc (the line-sum's node table) is what tools/gen_cheb.py writes, and the row-level interpolation keeps a
    Lorentzian wing to 2.3e-7 of itself at the distance the kernel uses it."""
    import re

    inc = open(os.path.join(ROOT, "radtxfr_amd", "csrc", "cheb8_64.inc")).read()
    off = np.array([float(v.rstrip("f")) for v in re.search(r"CHEB_OFF\[8\] = \{([^}]*)\}", inc).group(1).split(",")])
    k = np.arange(8)
    nodes = 31.5 + 32.0 * np.cos((2 * k + 1) * np.pi / 16)
    assert np.allclose(off, nodes, rtol=1e-7)

    def lagr(nd, xs):
        W = np.ones((len(xs), len(nd)))
        for j in range(len(nd)):
            for m in range(len(nd)):
                if m != j:
                    W[:, j] *= (xs - nd[m]) / (nd[j] - nd[m])
        return W

    p = np.arange(64.0)
    W = lagr(nodes, p)
    # row level: centre >= 2 full rows + 1 point before the row start (RTX_SC_NEAR = 2 since round 2; 3 gave 1.4e-8)
    src = open(os.path.join(ROOT, "radtxfr_amd", "csrc", "rtx_voigt_scatter.hip")).read()
    assert re.search(r"#define RTX_SC_NEAR 2\b", src)
    worst = 0.0
    for g in (0.5, 5.0, 30.0, 100.0):
        for d0 in (129.0, 161.0, 192.0):
            f = lambda t: 1.0 / ((t + d0) ** 2 + g * g)
            worst = max(worst, np.max(np.abs(W @ f(nodes) - f(p)) / f(p)))
    assert worst < 3e-7, worst
    # tile level (round 3): cheb16_tile.inc is what the generator writes; 16 nodes over a 16-row tile keep a wing whose centre
    # lies >= RTX_SC_TILE_DIST points outside the tile to 4.1e-8 of itself, ALSO after the carry to the 128 row nodes and
    # the row-level interpolation from there (the path the kernel takes), in the fp32 node positions the kernel uses
    inc16 = open(os.path.join(ROOT, "radtxfr_amd", "csrc", "cheb16_tile.inc")).read()
    toff = np.array([float(v.rstrip("f")) for v in re.search(r"TCHEB_OFF\[8\] = \{([^}]*)\}", inc16).group(1).split(",")])
    t = np.arange(16)
    tn = 511.5 + 512.0 * np.cos((2 * t + 1) * np.pi / 32)
    assert np.allclose(toff, tn[:8], rtol=1e-7)
    tn32 = tn.astype(np.float32)
    tn32[15 - np.arange(8)] = np.float32(1023.0) - tn32[:8]
    tnu = tn32.astype(np.float64)
    rows = re.findall(r"^  \{([^}]*)\},$", inc16[inc16.index("TCHEB_M"):], flags=re.M)
    M = np.array([[float(v.rstrip("f")) for v in r.split(",")] for r in rows])  # [16 tile nodes][128 row nodes]
    rownodes = (64.0 * np.arange(16)[:, None] + off[None, :]).ravel()
    assert M.shape == (16, 128) and np.allclose(M, lagr(tnu, rownodes).T, atol=2e-7)
    dist = int(re.search(r"#define RTX_SC_TILE_DIST (\d+)", src).group(1))
    assert dist == 512
    pt = np.arange(1024.0)
    worst = 0.0
    for g in (0.5, 10.0, 30.0, 100.0, 300.0):
        for u0 in (-float(dist), -700.0, 1023.0 + dist, 1023.0 + 900.0):
            f = lambda u: 1.0 / ((u - u0) ** 2 + g * g)
            at_rownodes = (f(tnu) @ M).reshape(16, 8)          # carry: tile nodes -> row nodes
            at_points = (W @ at_rownodes.T).T.ravel()          # row level: row nodes -> the 64 points of each row
            worst = max(worst, np.max(np.abs(at_points - f(pt)) / f(pt)))
    assert worst < 6e-8, worst


def test_window_taps_reproduce_the_reference_smoother():
    """engine.window_taps (the FIR handed to rtx_fir_reflect) against the oracle's smooth / symmetrised smooth,
    with the kernel's reflection rule emulated in NumPy."""
    from oracle import cpu_ref
    from radtxfr_amd import engine

    def fir(x, taps, c):
        n = len(x)
        k = np.arange(len(taps))
        out = np.zeros(n)
        for i in range(n):
            j = i + k - c
            j = np.where(j < 0, -j, j)
            j = np.where(j >= n, 2 * (n - 1) - j, j)
            out[i] = np.dot(taps, x[j])
        return out

    x = np.random.default_rng(7).standard_normal(257)
    for wl, win in ((11, "hanning"), (50, "hamming"), (7, "flat"), (20, "blackman"), (3, "bartlett")):
        t, c = engine.window_taps(wl, win)
        assert np.allclose(fir(x, t, c), cpu_ref.smooth(x, wl, win), rtol=0, atol=1e-14)
        t, c = engine.window_taps(wl, win, symmetric=True)
        assert len(t) % 2 == 1 and np.allclose(t, t[::-1])
        assert np.allclose(fir(x, t, c), cpu_ref.smooth_sym(x, wl, win), rtol=0, atol=1e-14)


def test_afit_xs_file_format(tmp_path):
    """AFIT_XS binary layout of misc/RT_gen_AbsXS_files.py:45-83 (b'v1', six float64, 128-byte description, float64
    data; default name XS-ID-TTTTK-PPPPPPPa.bin), checked against the same NumPy dtypes the reference serialises."""
    from radtxfr_amd import afit_xs

    X = np.linspace(400.0, 410.0, 4001)
    Y = np.random.default_rng(0).random(4001).astype(np.float32)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        fn = afit_xs.AFIT_XS_write(X, Y, 275.0, 101325 * 0.85, 1, "HITRAN2016 - HAPI - SDVoigt")
    finally:
        os.chdir(cwd)
    assert fn == "XS-01-0275K-086126Pa.bin"
    raw = open(os.path.join(tmp_path, fn), "rb").read()
    want = (np.array("v1", "<S2").tobytes() + np.array([X.min(), X.max(), X.size, 1, 275.0, 101325 * 0.85], "<f8").tobytes()
            + np.array("HITRAN2016 - HAPI - SDVoigt", "<S128").tobytes() + Y.astype("<f8").tobytes())
    assert raw == want and len(raw) == 2 + 48 + 128 + 8 * 4001
    back = afit_xs.AFIT_XS_read(os.path.join(tmp_path, fn))
    assert back["ID"] == 1 and back["T"] == 275.0 and back["n"] == 4001 and back["db"] == "HITRAN2016 - HAPI - SDVoigt"
    assert np.array_equal(back["Y"], Y.astype(np.float64)) and np.allclose(back["X"], X, rtol=0, atol=1e-12)


def test_reference_module_surface_is_complete():
    """Every public function of the reference's radiative_transfer.py exists in the shim module (names as listed in
    SURVEY section 2: the LBLRTM plumbing raises with an explanation, the reshape helpers behave as in the reference)."""
    from radtxfr_amd import radiative_transfer as rt

    for name in ("make_spectral_axis", "compute_TUD", "compute_OD", "planckian", "brightnessTemperature", "BT2L",
                 "compute_LWIR_apparent_radiance", "ILS_MAKO", "smooth", "reduceResolution", "rs1D", "rs2D", "rsND",
                 "write_tape5", "run_LBLRTM", "read_tape12"):
        assert callable(getattr(rt, name)), name
    a = np.arange(24).reshape(2, 3, 4)
    flat, dims = rt.rs1D(a)
    assert flat.shape == (24,) and dims == (2, 3, 4)
    m, dims = rt.rs2D(a)
    assert m.shape == (2, 12) and np.array_equal(rt.rsND(m, dims), a)
    assert rt.rs2D(np.arange(5))[0].shape == (1, 5) and rt.rs2D(3.0)[0].shape == (1, 1)
    for name in ("write_tape5", "run_LBLRTM", "read_tape12"):
        with pytest.raises(NotImplementedError):
            getattr(rt, name)()


def test_tud_downwelling_table_against_direct_sum():
    """The angle-summed transmission function G(S) = sum_q w_q exp(-S / cos theta_q) that rtx_tud tabulates (piecewise
    degree-6 polynomials), rebuilt here from the table the library hands out and the index rule its header
    states, against the direct sum over the reference's quadrature (radiative_transfer.py:368, 387): <= 1e-13 G(0) in absolute
    terms everywhere on [0, 48], <= 1e-9 relative where G has dropped by orders of magnitude. Host code only: no GPU."""
    import ctypes as C
    from radtxfr_amd import _lib
    lib = _lib.load()
    n = lib.rtx_tud_gtable_size()
    assert n % 8 == 0
    rng = np.random.default_rng(20261014)
    S = np.concatenate([rng.uniform(0, 1, 50000) ** 4, rng.uniform(0, 16, 50000), rng.uniform(16, 48, 20000), 10.0 ** rng.uniform(-12, -3, 20000),
                        [0.0, 1.0 / 64 - 1e-12, 1.0 / 64, 16.0 - 1e-9, 16.0, 47.999999]])
    for n_angle in (2, 3, 30, 96):
        tab = np.zeros(n)
        g0 = C.c_double(0.0)
        assert lib.rtx_tud_gtable(n_angle, tab.ctypes.data_as(C.c_void_p), C.byref(g0)) == 0
        tab = tab.reshape(-1, 8)
        th = np.arange(n_angle) * (np.pi / 2 / n_angle)   # np.linspace(0, pi/2, n, endpoint=False)
        w, sec = np.cos(th) * np.sin(th), 1.0 / np.cos(th)
        assert abs(g0.value - w.sum()) <= 1e-14 * w.sum()
        direct = (w[None, 1:] * np.exp(-S[:, None] * sec[None, 1:])).sum(1)
        # the kernel's index: float bits of (float(S) + 2^-6) below 16, half-unit intervals above
        sf = S.astype(np.float32)
        bits = (sf + np.float32(2.0 ** -6)).view(np.int32).astype(np.int64)
        base = int(np.float32(2.0 ** -6).view(np.int32)) >> 19
        n_log = (int(np.float32(16.0).view(np.int32)) >> 19) - base + 1
        idx = np.where(sf < 16.0, (bits >> 19) - base, n_log + ((sf - np.float32(16.0)) * np.float32(2.0)).astype(np.int64))
        idx = np.clip(idx, 0, tab.shape[0] - 1)
        u = S - tab[idx, 0]
        val = tab[idx, 7]
        for k in range(6, 0, -1):
            val = val * u + tab[idx, k]
        err = np.abs(val - direct)
        assert err.max() <= 1e-13 * g0.value, (n_angle, err.max())
        far = (direct < 1e-4 * g0.value) & (direct > 1e-14 * g0.value)   # below that nothing reaches fp32 output
        assert np.max(err[far] / direct[far]) <= 2e-8, (n_angle, np.max(err[far] / direct[far]))


def test_column_signature_is_nan_safe_and_sees_edits():
    """a column holding NaN (a blank field of a .par record) must not make the cached device table look stale on
    every call, yet every in-place edit -- of a finite value, or of which rows hold NaN -- must change the fingerprint;
    storage2cache_from_columns keeps numeric columns as ndarrays."""
    from radtxfr_amd import hapi
    a = np.linspace(1.0, 2.0, 1000)
    assert hapi._column_signature(a, 1000) == hapi._column_signature(a.copy(), 1000)
    b = a.copy()
    b[7] = np.nan
    s1 = hapi._column_signature(b, 1000)
    assert s1 == hapi._column_signature(b.copy(), 1000) and s1 != hapi._column_signature(a, 1000)
    c = b.copy()
    c[8] *= 1.0000001
    d = a.copy()
    d[9] = np.nan
    assert hapi._column_signature(c, 1000) != s1 and hapi._column_signature(d, 1000) != s1
    hapi.storage2cache_from_columns("sigtest", {"nu": a, "molec_id": np.ones(1000, dtype=np.int64), "global_upper_quanta": ["x"] * 1000})
    data = hapi.LOCAL_TABLE_CACHE.pop("sigtest")["data"]
    assert isinstance(data["nu"], np.ndarray) and isinstance(data["molec_id"], np.ndarray) and isinstance(data["global_upper_quanta"], list)
    assert data["nu"] is not a


def test_outer_band_series_matches_weideman_for_doppler_lines():
    """csrc/rtx_voigt_math.h: asymK_re<12>, the fp32 12-term asymptotic series the line-sum uses on the outer band rows
    (every lane |x| >= 5.5) of Doppler-dominated (y < 1) lines instead of the fp64 Weideman polynomial. Restated in NumPy
    float32 and compared with the oracle's hum1_wei (the reference's Weideman-24 value) in the parity metric -- error over
    max(value, 1e-3 of the line's peak, w(0) = 1) -- over 1e-5 <= y < 1 and 5.5 <= |x| < 15."""
    from oracle import cpu_ref as ref
    f = np.float32

    def asym12(x, y):
        x = x.astype(f)
        y = f(y)
        r2 = x * x + y * y
        inv = f(1) / r2
        zr, zi = x * inv, -y * inv
        ur, ui = zr * zr - zi * zi, f(2) * zr * zi
        c = [1.0]
        for k in range(1, 12):
            c.append(c[-1] * (2 * k - 1) / 2.0)
        pr, pi = np.full_like(x, f(c[11])), np.zeros_like(x)
        for k in range(10, -1, -1):
            pr, pi = pr * ur - pi * ui + f(c[k]), pr * ui + pi * ur
        return (-f(0.5641895835477563) * (zr * pi + zi * pr)).astype(np.float64)

    worst = 0.0
    for y in (1e-5, 1e-4, 1e-3, 1e-2, 0.05, 0.2, 0.5, 0.99):
        for sgn in (1.0, -1.0):
            x = sgn * np.linspace(5.5, 15.0, 6001)
            x = x[np.abs(x) + y < 15.0]
            w = ref.hum1_wei(x, np.full_like(x, y))[0]
            worst = max(worst, float(np.max(np.abs(asym12(x, y) - w) / np.maximum(w, 1e-3))))
    assert worst <= 1e-6, worst


def test_range_search_step_count():
    """csrc/rtx_voigt.hip: bound_step / tile_ranges_kernel. The 17-ary search of 16 lanes (16 probes per step; the bracket
    shrinks to at most its stride) restated in Python: with n_steps = 1 + the number of times n can be divided by 17 before
    reaching 0 -- what launch_tile_ranges passes -- it returns bisect_left / bisect_right for every size class."""
    import bisect
    G = 16

    def search(ic, v, strict, n_steps):
        lo, hi = 0, len(ic)
        for _ in range(n_steps):
            width = hi - lo
            if width <= 0:
                continue
            stride = (width + G) // (G + 1)
            c = 0
            for sub in range(G):
                p = lo + (sub + 1) * stride - 1
                if p < hi and (ic[p] < v if strict else ic[p] <= v):
                    c += 1
            cap = lo + (c + 1) * stride - 1
            if c < G and cap < hi:
                hi = cap
            lo += c * stride
        return lo

    rng = np.random.default_rng(1)
    for n in (1, 2, 16, 17, 18, 288, 289, 290, 4913, 4914, 20000):
        steps, w = 1, n
        while w > 0:
            w //= 17
            steps += 1
        for _ in range(20):
            ic = np.sort(rng.integers(-50, 3 * n + 50, n)).tolist()
            for v in rng.integers(-60, 3 * n + 60, 6).tolist() + [ic[0], ic[-1], ic[n // 2]]:
                assert search(ic, v, True, steps) == bisect.bisect_left(ic, v)
                assert search(ic, v, False, steps) == bisect.bisect_right(ic, v)


def test_sdvoigt_regime_zones_and_node_error_bounds():
    """The closed forms behind the speed-dependent Voigt kernel's node levels (csrc/rtx_sdvoigt.hip: sd_zones), restated in
    NumPy. (1) pcqsdhc's PART4 takes hum1_wei's one-term asymptote for both arguments iff |x1| + y1 >= 15 with
    Z1 = sqrt(X + Y) - csqrtY = y1 - i x1; with sqrt(X + Y) = p + iq that is p + |q| >= K = 15 + csqrtY, and
    (p + |q|)^2 = |X + Y| + |Im X| gives |Im X| >= (K^4 - R^2) / (2 K^2), R = Re X + Y: the threshold is checked against the
    pointwise test on both sides, over the physical range of the records. (2) The PART2 / PART3 thresholds are circles in X.
    (3) Interpolating the far-wing closed form at 32 Chebyshev nodes over 1024 points (centre >= 256 points away) and at 12
    nodes over 64 points (centre >= 3 rows away) keeps 1e-10 of the line's own contribution, the figures the kernel's header
    quotes."""
    rng = np.random.default_rng(17)

    def far_threshold(gam0, gam2, cte):
        a, c2 = gam0 - 1.5 * gam2, gam2
        sY = 1.0 / (2.0 * cte * c2)
        Y = sY * sY
        K2 = (15.0 + sY) ** 2
        R = a / c2 + Y
        d = 225.0 + 30.0 * sY - a / c2
        return c2 * (d * (K2 + R)) / (2.0 * K2) if d > 0 else 0.0

    def is_far(gam0, gam2, cte, delta):
        a, c2 = gam0 - 1.5 * gam2, gam2
        sY = 1.0 / (2.0 * cte * c2)
        X = complex(a, delta) / c2
        Z1 = np.sqrt(X + sY * sY) - sY
        return abs(Z1.imag) + Z1.real >= 15.0

    n_checked = 0
    for _ in range(400):
        gam0 = 10.0 ** rng.uniform(-6, -0.5)
        gam2 = gam0 * 10.0 ** rng.uniform(-6, -0.2) * 0.6
        cte = np.sqrt(np.log(2.0)) / 10.0 ** rng.uniform(-4, -1.5)
        t = far_threshold(gam0, gam2, cte)
        if t <= 0.0:
            assert is_far(gam0, gam2, cte, 0.0)
            continue
        # (probed at +-1e-6: for a tiny Gamma2 the pointwise test itself is only good to ~1e-8 of delta -- Z1 is the difference
        # of two numbers of size csqrtY, and PART4 is only entered where |X| > 3e-8 Y, which bounds that noise by 1.3e-8 delta;
        # the kernel keeps two grid points between a switch and the rows it takes to the node levels)
        assert is_far(gam0, gam2, cte, t * (1 + 1e-6)) and not is_far(gam0, gam2, cte, t * (1 - 1e-6)), (gam0, gam2, cte, t)
        n_checked += 1
        # PART2: |X| <= 3e-8 Y  <=>  delta <= sqrt((3e-8 Y c2)^2 - a^2); PART3: |X| >= 1e15 Y likewise
        a, c2 = gam0 - 1.5 * gam2, gam2
        Y = (1.0 / (2.0 * cte * c2)) ** 2
        for fac in (3.0e-8, 1.0e15):
            tt = fac * Y * c2
            if tt > a and np.isfinite(tt * tt):
                dth = np.sqrt(tt * tt - a * a)
                if dth > 1e-3 * a:  # (the square root loses the threshold's digits when it is a small difference)
                    for s, want in ((1 + 1e-6, False), (1 - 1e-6, True)):
                        assert (abs(complex(a, dth * s) / c2) <= fac * Y) == want
    assert n_checked > 200

    # (3) node levels on the far-wing closed form f(t1) - f(t2), t = Z, for a narrow line (singularities ~ on the real axis)
    def far_profile(delta, gam0=0.07, gam2=0.009, cte=574.0):
        a, c2 = gam0 - 1.5 * gam2, gam2
        sY = 1.0 / (2.0 * cte * c2)
        X = (a + 1j * delta) / c2
        t1 = np.sqrt(X + sY * sY) - sY
        t2 = t1 + 2.0 * sY
        return (cte * (-2.0 * sY) * (0.5 - t1 * t2) / ((0.5 + t1 * t1) * (0.5 + t2 * t2))).real

    def cheb_err(n_nodes, L, dist, step):
        h = (L - 1) / 2.0
        xj = h + h * np.cos((2 * np.arange(n_nodes) + 1) * np.pi / (2 * n_nodes))
        p = np.arange(L, dtype=np.float64)
        fj = far_profile((xj + dist) * step)
        w = np.ones((n_nodes, L))
        for j in range(n_nodes):
            for m in range(n_nodes):
                if m != j:
                    w[j] *= (p - xj[m]) / (xj[j] - xj[m])
        want = far_profile((p + dist) * step)
        return float(np.max(np.abs(fj @ w - want) / want))

    step = 0.0025
    # the far regime of this line starts ~400 points from its centre; tile level from max(256 points, that)
    assert cheb_err(32, 1024, 420.0, step) <= 1e-10
    assert cheb_err(32, 1024, 256.0, 0.05) <= 1e-10   # coarse grid: the regime boundary lies inside 256 points
    assert cheb_err(12, 64, 420.0, step) <= 1e-10
    assert cheb_err(12, 64, 129.0, 0.05) <= 1e-10     # three rows from the centre row at the least: >= 129 points to the centre
