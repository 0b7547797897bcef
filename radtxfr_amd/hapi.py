"""Drop-in for the hot-path part of the reference's vendored HITRAN API (misc/hapi.py).

Only the functions on the north-star path are provided, with the reference's names, argument
meaning and error behaviour:

    absorptionCoefficient_Voigt   misc/hapi.py:10906-11141   -> HIP prologue + line-sum kernels
    LOCAL_TABLE_CACHE             misc/hapi.py:438-463       (same dict layout)
    PYTIPS / partitionSum pieces  misc/hapi.py:9568-9582, 10030
    abundance, molecularMass      misc/hapi.py:5088-5124
    volumeConcentration           misc/hapi.py:10163-10164

The database client (fetch/select/...), the other line profiles and the slit functions are out of
scope. There is no CPU fallback: without the HIP library and a
GPU, absorptionCoefficient_Voigt raises.
"""
import math

import numpy as np
import torch

from . import engine
from .tips import PYTIPS, abundance, molecularMass, known_isotopologues  # noqa: F401  (re-exported API)

volumeConcentration = engine.volumeConcentration

DefaultIntensityThreshold = 0.0  # misc/hapi.py:10214
DefaultOmegaWingHW = 50.0        # misc/hapi.py:10218

# name -> {'header': {'number_of_rows': n, ...}, 'data': {column: list-or-array}}
LOCAL_TABLE_CACHE = {}
_DEVICE_TABLES = {}   # tuple(table names) -> (signature, engine.LineTable); least recently used entries are closed
_DEVICE_TABLES_MAX = 8


def storage2cache_from_columns(TableName, columns):
    """Convenience: register a column dict as a table (what db_begin()/fetch() leave in the cache)."""
    n = len(columns["nu"])
    # numeric columns are kept as ndarrays (as hitran_par.storage2cache does): the reference's Python lists would cost a
    # list -> array conversion of every column on every call (~50 ms for 100 000 rows, ten times the device time)
    def keep(v):
        a = np.asarray(v)
        return a.copy() if a.dtype.kind in "fiub" else list(v)
    LOCAL_TABLE_CACHE[TableName] = {"header": {"number_of_rows": n, "table_name": TableName},
                                    "data": {k: keep(v) for k, v in columns.items()}}


_USED_COLS = ("molec_id", "local_iso_id", "nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air",
              "n_self", "deltap_air", "delta_self", "deltap_self", "SD_air", "SD_self")
_SIG_WEIGHTS = {}


def _column_signature(v, nrow):
    """Content fingerprint of one column (list or array): length + a dot product with fixed pseudo-random weights, so
    that any in-place edit (a rescaled sw, a tweaked gamma_air, two rows swapped) changes it. ~30 us per 100 000-row
    ndarray column; list columns (the reference's layout) pay the list -> array conversion."""
    a = np.asarray(v)[:nrow]
    if a.dtype.kind not in "fiub":
        a = a.astype(np.float64)
    w = _SIG_WEIGHTS.get(a.size)
    if w is None:
        if len(_SIG_WEIGHTS) > 16:
            _SIG_WEIGHTS.clear()
        w = _SIG_WEIGHTS[a.size] = np.random.default_rng(12345).uniform(0.5, 1.5, a.size)
    if not a.size:
        return (0, 0.0)
    # (np.einsum, not np.dot: the BLAS dot product is multi-threaded -- 64 OpenBLAS threads on a box whose share is 16 CPUs --
    # and every few calls one of the 15 products stalled for 70-80 ms waking them: the sporadic slow drop-in calls of
    # profiles/r3_time_dropin.txt with a table name / column dict, and every other afit_xs.cross_section_grid call)
    if a.dtype != np.float64:
        a = a.astype(np.float64)
    d = float(np.einsum("i,i->", a, w))
    if d != d:
        # a NaN in the column (e.g. a blank field of a .par record): NaN never compares equal, which would rebuild the
        # device table on every call. Fingerprint the finite part and the positions of the NaNs instead.
        bad = np.isnan(a.astype(np.float64, copy=False))
        return (a.size, float(np.einsum("i,i->", np.where(bad, 0.0, a), w)), float(np.einsum("i,i->", bad.astype(np.float64), w)))
    return (a.size, d)


def _device_table(names):
    """Device LineTable for one or several cached tables (concatenated). The reference re-reads LOCAL_TABLE_CACHE on
    every call (misc/hapi.py:11044-11125); here the device copy is reused only while a content fingerprint of EVERY
    uploaded column is unchanged, so in-place edits of a cached table are seen (tests/test_gpu_parity.py)."""
    key = tuple(names)
    sig = []
    for n in names:
        d, nrow = LOCAL_TABLE_CACHE[n]["data"], LOCAL_TABLE_CACHE[n]["header"]["number_of_rows"]
        sig.append((nrow,) + tuple((k, _column_signature(d[k], nrow)) for k in _USED_COLS if k in d))
    sig = tuple(sig)
    hit = _DEVICE_TABLES.get(key)
    if hit is not None and hit[0] == sig:
        _DEVICE_TABLES[key] = _DEVICE_TABLES.pop(key)  # mark most recently used
        return hit[1]
    cols = {}
    keys = None
    for n in names:
        d = LOCAL_TABLE_CACHE[n]["data"]
        nrow = LOCAL_TABLE_CACHE[n]["header"]["number_of_rows"]
        keys = set(d.keys()) if keys is None else keys & set(d.keys())
        for k, v in d.items():
            cols.setdefault(k, []).append(np.asarray(v)[:nrow])
    use = [k for k in _USED_COLS if k in keys]
    for req in ("molec_id", "local_iso_id", "nu", "sw", "elower", "gamma_air", "n_air", "delta_air"):
        if req not in use:
            raise Exception("table(s) %s lack the column %s" % (names, req))
    merged = {k: np.concatenate(cols[k]) for k in use}
    if "gamma_self" not in merged:
        merged["gamma_self"] = np.zeros(merged["nu"].size)  # hapi: missing gamma_<species> -> 0 (:11097-11100)
    if hit is not None:
        hit[1].close()
    tbl = engine.LineTable(merged)
    _DEVICE_TABLES.pop(key, None)
    _DEVICE_TABLES[key] = (sig, tbl)
    while len(_DEVICE_TABLES) > _DEVICE_TABLES_MAX:  # bound the device memory held for callers' tables
        old_key = next(iter(_DEVICE_TABLES))
        _DEVICE_TABLES.pop(old_key)[1].close()
        for n in old_key:
            if n.startswith("__dict_"):
                LOCAL_TABLE_CACHE.pop(n, None)
    return tbl


def listOfTuples(a):
    """misc/hapi.py:10221-10228."""
    if type(a) not in set([list, tuple]):
        a = [a]
    return a


def arange_(lower, upper, step):
    """misc/hapi.py:133-139 with the float-`num` crash fixed (npnt cast to int)."""
    npnt = math.floor((upper - lower) / step) + 1
    upper_new = lower + step * (npnt - 1)
    if abs((upper - upper_new) - step) < 1e-10:
        upper_new += step
        npnt += 1
    return np.linspace(lower, upper_new, int(npnt))


def _absorption_coefficient(profile, Components, SourceTables, partitionFunction, Environment, OmegaRange, OmegaStep, OmegaWing,
                            IntensityThreshold, OmegaWingHW, GammaL, HITRAN_units, LineShift, File, Format, OmegaGrid,
                            WavenumberRange, WavenumberStep, WavenumberWing, WavenumberWingHW, WavenumberGrid, Diluent,
                            EnvDependences):
    """Common body of absorptionCoefficient_Voigt / _Lorentz / _Doppler (they share getDefaultValuesForXsect, the
    abundance bookkeeping and the per-line prologue; misc/hapi.py:10906-11141, 11144-11375, 11384-11559)."""
    if WavenumberRange is not None: OmegaRange = WavenumberRange
    if WavenumberStep: OmegaStep = WavenumberStep
    if WavenumberWing: OmegaWing = WavenumberWing
    if WavenumberWingHW: OmegaWingHW = WavenumberWingHW
    if WavenumberGrid is not None: OmegaGrid = WavenumberGrid
    if EnvDependences:
        raise NotImplementedError("EnvDependences hooks are not supported by the HIP line-sum")
    Components = listOfTuples(Components)
    SourceTables = listOfTuples(SourceTables)
    # getDefaultValuesForXsect, misc/hapi.py:10231-10281
    if SourceTables[0] is None:
        SourceTables = ["__BUFFER__"]
    for TableName in SourceTables:
        if TableName not in LOCAL_TABLE_CACHE:
            raise Exception("%s: no such table. Check tableList() for more info." % TableName)
    if Environment is None:
        Environment = {"T": 296.0, "p": 1.0}
    tbl = _device_table(SourceTables)
    if Components == [None]:
        Components = [p for p in tbl.species if p != (0, 0)]
    if OmegaRange is None:
        nu = tbl.cols["nu"]
        OmegaRange = (float(nu.min()), float(nu.max())) if nu.size else (0.0, 0.0)
    if OmegaStep is None:
        OmegaStep = 0.01
    if OmegaWing is None:
        OmegaWing = 0.0
    if not Format:
        Format = "%.12f %e"
    if OmegaStep > (0.005 if profile == 2 else 0.1):  # :11027 / :11452
        print("WARNING: Big wavenumber step: possible accuracy decline")
    if OmegaGrid is not None:
        Omegas = np.sort(np.asarray(OmegaGrid, dtype=np.float64))
    else:
        Omegas = arange_(OmegaRange[0], OmegaRange[1], OmegaStep)
    T = Environment["T"]
    p = Environment["p"]
    # abundances, misc/hapi.py:10996-11009
    ABUNDANCES, NATURAL = {}, {}
    for Component in Components:
        M, I = int(Component[0]), int(Component[1])
        nat = abundance(M, I)
        ABUNDANCES[(M, I)] = Component[2] if len(Component) >= 3 else nat
        NATURAL[(M, I)] = nat
    factor = 1.0 if HITRAN_units else volumeConcentration(p, T)
    GammaL = GammaL.lower()
    if profile == 2:
        Diluent = {"air": 1.0 if LineShift else 0.0}  # Doppler: Shift0 = delta_air*p, or none (misc/hapi.py:11510-11513)
    if not Diluent:
        if GammaL == "gamma_air":
            Diluent = {"air": 1.0}
        elif GammaL == "gamma_self":
            Diluent = {"self": 1.0}
        else:
            raise Exception("Unknown GammaL value: %s" % GammaL)
    dil = {k.lower(): float(v) for k, v in Diluent.items()}
    extra = set(dil) - {"air", "self"}
    if extra:
        raise NotImplementedError("diluents %s are not supported (air, self only)" % sorted(extra))
    if Omegas.size < 2:
        raise NotImplementedError("the HIP line-sum needs a grid of at least 2 points")
    grid = engine.Grid.from_axis(Omegas)
    # per-species weight = factor / natural * abundance (misc/hapi.py:11136-11137); 0 filters the species out (:11066)
    w = np.zeros((len(tbl.species), 1))
    for s, mi in enumerate(tbl.species):
        if mi in ABUNDANCES:
            w[s, 0] = factor / NATURAL[mi] * ABUNDANCES[mi]
    # fold a power of two into the fp32 strengths so HITRAN-unit intensities (~1e-19..1e-30) stay normal
    smax = float(np.max(tbl.cols["sw"])) * float(np.max(w)) if tbl.n and np.max(w) > 0 else 1.0
    scale = 2.0 ** (-math.floor(math.log2(smax))) if smax > 0 and math.isfinite(smax) else 1.0
    if tbl.n == 0:
        Xsect = np.zeros(Omegas.size)
    else:
        out = torch.empty((1, grid.n), dtype=torch.float64, device=engine.device())
        engine.voigt_sum(tbl, grid, [T], [p], w, out_f64=out, dil_air=dil.get("air", 0.0), dil_self=dil.get("self", 0.0),
                         omega_wing=OmegaWing, omega_wing_hw=OmegaWingHW, intensity_threshold=IntensityThreshold,
                         scale=scale, partitionFunction=partitionFunction, profile=profile)
        Xsect = out[0].cpu().numpy()
    if File:
        with open(File, "w") as f:
            for o, x in zip(Omegas, Xsect):
                f.write((Format % (o, x)) + "\n")
    return Omegas, Xsect


def absorptionCoefficient_Voigt(Components=None, SourceTables=None, partitionFunction=PYTIPS, Environment=None,
                                OmegaRange=None, OmegaStep=None, OmegaWing=None,
                                IntensityThreshold=DefaultIntensityThreshold, OmegaWingHW=DefaultOmegaWingHW,
                                GammaL="gamma_air", HITRAN_units=True, LineShift=True, File=None, Format=None,
                                OmegaGrid=None, WavenumberRange=None, WavenumberStep=None, WavenumberWing=None,
                                WavenumberWingHW=None, WavenumberGrid=None, Diluent={}, EnvDependences=None):
    """Absorption coefficient with the Voigt profile; same inputs/outputs as misc/hapi.py:10906-11141.

    Returns (Omegas, Xsect) as float64 NumPy arrays. The sum over lines runs on the GPU
    (rtx_line_prep + rtx_voigt_sum); line strengths are carried in fp32 with a power-of-two scale,
    so Xsect agrees with the reference to ~1e-6 relative, not bit for bit.
    Not supported (raises): EnvDependences hooks, diluents other than air/self, non-uniform grids.
    """
    return _absorption_coefficient(0, Components, SourceTables, partitionFunction, Environment, OmegaRange, OmegaStep, OmegaWing,
                                   IntensityThreshold, OmegaWingHW, GammaL, HITRAN_units, LineShift, File, Format, OmegaGrid,
                                   WavenumberRange, WavenumberStep, WavenumberWing, WavenumberWingHW, WavenumberGrid, Diluent,
                                   EnvDependences)


def absorptionCoefficient_Lorentz(Components=None, SourceTables=None, partitionFunction=PYTIPS, Environment=None,
                                  OmegaRange=None, OmegaStep=None, OmegaWing=None,
                                  IntensityThreshold=DefaultIntensityThreshold, OmegaWingHW=DefaultOmegaWingHW,
                                  GammaL="gamma_air", HITRAN_units=True, LineShift=True, File=None, Format=None,
                                  OmegaGrid=None, WavenumberRange=None, WavenumberStep=None, WavenumberWing=None,
                                  WavenumberWingHW=None, WavenumberGrid=None, Diluent={}, EnvDependences=None):
    """Absorption coefficient with the Lorentz profile; same inputs/outputs as misc/hapi.py:11144-11375
    (PROFILE_LORENTZ :10150, wing max(OmegaWing, OmegaWingHW*Gamma0) :11364). GPU path and limits as for Voigt."""
    return _absorption_coefficient(1, Components, SourceTables, partitionFunction, Environment, OmegaRange, OmegaStep, OmegaWing,
                                   IntensityThreshold, OmegaWingHW, GammaL, HITRAN_units, LineShift, File, Format, OmegaGrid,
                                   WavenumberRange, WavenumberStep, WavenumberWing, WavenumberWingHW, WavenumberGrid, Diluent,
                                   EnvDependences)


def absorptionCoefficient_Doppler(Components=None, SourceTables=None, partitionFunction=PYTIPS, Environment=None,
                                  OmegaRange=None, OmegaStep=None, OmegaWing=None,
                                  IntensityThreshold=DefaultIntensityThreshold, OmegaWingHW=DefaultOmegaWingHW,
                                  ParameterBindings=None, EnvironmentDependencyBindings=None,
                                  GammaL="dummy", HITRAN_units=True, LineShift=True, File=None, Format=None,
                                  OmegaGrid=None, WavenumberRange=None, WavenumberStep=None, WavenumberWing=None,
                                  WavenumberWingHW=None, WavenumberGrid=None):
    """Absorption coefficient with the Doppler (Gauss) profile; same inputs/outputs as misc/hapi.py:11384-11559
    (PROFILE_DOPPLER :10160, its own GammaD constants :11534-11538, wing max(OmegaWing, OmegaWingHW*GammaD) :11540,
    shift delta_air*p only when LineShift). ParameterBindings / EnvironmentDependencyBindings are accepted and ignored,
    as in the reference. Values below exp(-225) of a line's peak (|nu - nu0| > 18 GammaD) come out 0."""
    return _absorption_coefficient(2, Components, SourceTables, partitionFunction, Environment, OmegaRange, OmegaStep, OmegaWing,
                                   IntensityThreshold, OmegaWingHW, "gamma_air", HITRAN_units, LineShift, File, Format, OmegaGrid,
                                   WavenumberRange, WavenumberStep, WavenumberWing, WavenumberWingHW, WavenumberGrid, {}, None)


absorptionCoefficient_Gauss = absorptionCoefficient_Doppler  # misc/hapi.py:11561


def absorptionCoefficient_SDVoigt(Components=None, SourceTables=None, partitionFunction=PYTIPS, Environment=None,
                                  OmegaRange=None, OmegaStep=None, OmegaWing=None,
                                  IntensityThreshold=DefaultIntensityThreshold, OmegaWingHW=DefaultOmegaWingHW,
                                  GammaL="gamma_air", HITRAN_units=True, LineShift=True, File=None, Format=None,
                                  OmegaGrid=None, WavenumberRange=None, WavenumberStep=None, WavenumberWing=None,
                                  WavenumberWingHW=None, WavenumberGrid=None, Diluent={}, EnvDependences=None):
    """Speed-dependent Voigt, signature of misc/hapi.py:10657-10904.

    Tables without speed-dependence columns (the 160-character HITRAN .par format has none) give Gamma2 = 0, for which
    pcqsdhc takes its PART1 branch (:9908-9915), i.e. the Voigt profile: those go through the fp32 Voigt line-sum.
    Tables with non-zero SD_air / SD_self (:10884-10890) go through rtx_sdvoigt_sum: pcqsdhc PART2-4 in fp64, far wings at
    Chebyshev nodes (the path of the reference's cross-section generator, misc/RT_gen_AbsXS_files.py:90)."""
    sd = False
    for name in listOfTuples(SourceTables):
        if name is None or name not in LOCAL_TABLE_CACHE:
            continue
        data = LOCAL_TABLE_CACHE[name]["data"]
        for col in ("SD_air", "SD_self"):
            if col in data and np.any(np.asarray(data[col], dtype=np.float64) != 0.0):
                sd = True
    return _absorption_coefficient(3 if sd else 0, Components, SourceTables, partitionFunction, Environment, OmegaRange, OmegaStep,
                                   OmegaWing, IntensityThreshold, OmegaWingHW, GammaL, HITRAN_units, LineShift, File, Format,
                                   OmegaGrid, WavenumberRange, WavenumberStep, WavenumberWing, WavenumberWingHW, WavenumberGrid,
                                   Diluent, EnvDependences)


_HT_PREFIXES = ("gamma_HT_", "n_HT_", "delta_HT_", "deltap_HT_", "nu_HT_", "kappa_HT_", "eta_HT_")


def absorptionCoefficient_HT(Components=None, SourceTables=None, partitionFunction=PYTIPS, Environment=None,
                             OmegaRange=None, OmegaStep=None, OmegaWing=None,
                             IntensityThreshold=DefaultIntensityThreshold, OmegaWingHW=DefaultOmegaWingHW,
                             GammaL="gamma_air", HITRAN_units=True, LineShift=True, File=None, Format=None,
                             OmegaGrid=None, WavenumberRange=None, WavenumberStep=None, WavenumberWing=None,
                             WavenumberWingHW=None, WavenumberGrid=None, Diluent={}, EnvDependences=None):
    """Hartmann-Tran profile, signature of misc/hapi.py:10302-10653 -- for tables WITHOUT Hartmann-Tran columns.

    The reference looks each parameter up under its HT name first (gamma_HT_0_<species>_<Tref>, n_HT_..., delta_HT_...,
    nu_HT_..., eta_HT_..., :10505-10640) and falls back to the Voigt-style columns; a table that has none of the HT
    names gives nuVC = eta = 0, Gamma2 from SD_<species> (:10590-10599), i.e. exactly absorptionCoefficient_SDVoigt
    (checked against the reference in the build container: 6e-16). That case is evaluated here; non-zero HT columns raise
    NotImplementedError (velocity-changing collisions and correlation are not implemented on the GPU)."""
    for name in listOfTuples(SourceTables):
        if name is None or name not in LOCAL_TABLE_CACHE:
            continue
        for col, vals in LOCAL_TABLE_CACHE[name]["data"].items():
            if col.startswith(_HT_PREFIXES) and np.any(np.asarray(vals, dtype=np.float64) != 0.0):
                raise NotImplementedError("absorptionCoefficient_HT: table %r has the Hartmann-Tran column %s; only the "
                                          "Voigt / speed-dependent Voigt limits are implemented" % (name, col))
    return absorptionCoefficient_SDVoigt(Components, SourceTables, partitionFunction, Environment, OmegaRange, OmegaStep, OmegaWing,
                                         IntensityThreshold, OmegaWingHW, GammaL, HITRAN_units, LineShift, File, Format, OmegaGrid,
                                         WavenumberRange, WavenumberStep, WavenumberWing, WavenumberWingHW, WavenumberGrid, Diluent,
                                         EnvDependences)


absorptionCoefficient = absorptionCoefficient_HT  # the reference's profile selector alias, misc/hapi.py:11377


# ---- loading line files: what the reference's scripts do before the line-sum ---------------------------------------
# (misc/RT_gen_AbsXS_files.py:12: db_begin(folder)). hapi's SQL-like layer (select / sort / group, misc/hapi.py:433-3216)
# is out of scope; rows are filtered with NumPy on LOCAL_TABLE_CACHE[name]['data'] instead.
VARIABLES = {"BACKEND_DATABASE_NAME": "data"}


def db_begin(db=None):
    """Load the tables of folder `db` (default 'data') into LOCAL_TABLE_CACHE as misc/hapi.py:5205-5221 /
    loadCache :1718-1730 does: every `<name>.header` names a table whose rows are in `<name>.data` (else `<name>.par`);
    a `.par` file without a header is read with the default 160-character HITRAN layout. Parsing follows the
    reference's storage2cache row by row (radtxfr_amd/hitran_par.py). Returns the list of table names loaded."""
    import os

    from . import hitran_par

    folder = "data" if db is None else db
    os.makedirs(folder, exist_ok=True)
    VARIABLES["BACKEND_DATABASE_NAME"] = folder
    files = sorted(os.listdir(folder))
    names = [f[:-len(".header")] for f in files if f.endswith(".header")]
    names += [f[:-len(".par")] for f in files if f.endswith(".par") and f[:-len(".par")] not in names]
    for name in names:
        path = os.path.join(folder, name + ".data")
        if not os.path.isfile(path):
            path = os.path.join(folder, name + ".par")
            if not os.path.isfile(path):
                raise Exception('Lonely header "%s"' % path)
        hitran_par.storage2cache(name, path)
    return names


def tableList():
    """Names of the cached tables (misc/hapi.py:5168)."""
    return list(LOCAL_TABLE_CACHE.keys())
