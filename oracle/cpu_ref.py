"""CPU oracle: fp64 NumPy restatement of the reference's LWIR hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker. The product path (radtxfr_amd/) never imports it and
fails loudly when the HIP library is missing.

Parity pin: the reference has no tests or golden vectors of its own, so this
restatement is pinned by golden vectors captured from the imported reference
(oracle/make_golden.py -> tests/golden/*.npz; checked by tests/test_oracle_golden.py to
~1e-13 relative).

All citations are file:line under the upstream reference checkout.
"""
import bisect as _bisect
import os

import numpy as np

# ---------------------------------------------------------------------------------------
# radiative_transfer.py constants (:71-72)
C1 = 1.19104295315e-16  # [J m^2 / s]
C2 = 1.43877736830e-02  # [m K]

# misc/hapi.py constants (:84-92, :10171, :11085)
CBOLTS = 1.380648813e-16  # erg/K
CC = 2.99792458e10  # cm/s
CMASSMOL = 1.66053873e-27
HAPI_C2 = 1.4388028496642257  # cm K   (EnvironmentDependency_Intensity)
TREF = 296.0
PREF = 1.0

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "radtxfr_amd", "data")
_tips = None


def _tables():
    global _tips
    if _tips is None:
        d = np.load(os.path.join(_DATA, "tips2011.npz"))
        _tips = {
            "tdat": d["tdat"].astype(np.float64),
            "q": {tuple(k): d["q"][r] for r, k in enumerate(d["mi"].tolist())},  # float32 rows
            "abun": {tuple(k): float(v) for k, v in zip(d["iso_mi"].tolist(), d["iso_abun"])},
            "mass": {tuple(k): float(v) for k, v in zip(d["iso_mi"].tolist(), d["iso_mass"])},
        }
    return _tips


# ---------------------------------------------------------------------------------------
# Planck / spectral axis                      radiative_transfer.py:251-271, 792-848
def make_spectral_axis(Xmin, Xmax, DVOUT):
    """:269-270 with the float->int crash fixed: spacing is (Xmax-Xmin)/(nX-1)."""
    nX = int(np.ceil((Xmax - Xmin) / DVOUT))
    return np.linspace(Xmin, Xmax, nX)


def planckian(X_in, T_in, wavelength=False):
    """radiative_transfer.py:792-848.  Output (X.size, *T.shape), uW/(cm^2 sr cm^-1)."""
    X = np.asarray(np.copy(X_in), dtype=np.float64).flatten()
    T = np.asarray(np.copy(T_in), dtype=np.float64)
    X = X[:, np.newaxis]
    dimsT = T.shape
    T = T.flatten()[np.newaxis, :]
    if wavelength or np.mean(X) < 50:
        X = X * 1e-6
        L = C1 / (X ** 5 * (np.exp(C2 / (X * T)) - 1))
        L *= 1e-4
    else:
        X = X * 100
        L = C1 * X ** 3 / (np.exp(C2 * X / T) - 1)
        L *= 1e4
    return np.reshape(L, (X.size, *dimsT))


def brightnessTemperature(X_in, L_in, wavelength=False, bad_value=np.nan, spectral_dim=0):
    """radiative_transfer.py:851-933 (section 8f 'next' row 1)."""
    X = np.asarray(np.copy(X_in), dtype=np.float64).flatten()
    L = np.asarray(np.copy(L_in), dtype=np.float64)
    if spectral_dim != 0:
        L = np.swapaxes(L, 0, spectral_dim)
    X = X[:, np.newaxis]
    if L.ndim == 1:
        L = L[:, np.newaxis]
        dimsL = L.shape
    else:
        dimsL = L.shape
        L = L.reshape((dimsL[0], int(np.prod(dimsL[1:]))))
    with np.errstate(all="ignore"):
        if wavelength or np.mean(X) < 50:
            X = X * 1e-6
            L = L * 1e4
            T = C2 / (X * np.log(1 + C1 / (X ** 5 * L)))
        else:
            X = X * 100
            L = L * 1e-4
            T = C2 * X / np.log(C1 * X ** 3 / L + 1)
    ixBad = ~np.isfinite(L) | (L <= 0)
    T[ixBad] = bad_value
    if [*dimsL[1:]] != [1]:
        T = np.reshape(T, (X.size, *dimsL[1:]))
    if spectral_dim != 0:
        T = np.swapaxes(T, 0, spectral_dim)
    return T


def BT2L(X_in, T_in, wavelength=False, bad_value=np.nan, spectral_dim=0):
    """radiative_transfer.py:936-1014 (section 8f 'next' row 1)."""
    X = np.asarray(np.copy(X_in), dtype=np.float64).flatten()
    T = np.asarray(np.copy(T_in), dtype=np.float64)
    if spectral_dim != 0:
        T = np.swapaxes(T, 0, spectral_dim)
    X = X[:, np.newaxis]
    if T.ndim == 1:
        T = T[:, np.newaxis]
        dimsT = T.shape
    else:
        dimsT = T.shape
        T = T.reshape((dimsT[0], int(np.prod(dimsT[1:]))))
    with np.errstate(all="ignore"):
        if wavelength or np.mean(X) < 50:
            X = X * 1e-6
            L = C1 / (X ** 5 * (np.exp(C2 / (X * T)) - 1))
            L *= 1e-4
        else:
            X = X * 100
            L = C1 * X ** 3 / (np.exp(C2 * X / T) - 1)
            L *= 1e4
    ixBad = ~np.isfinite(L) | (T <= 0)
    L[ixBad] = bad_value
    L = np.reshape(L, (X.size, *dimsT[1:]))
    if spectral_dim != 0:
        L = np.swapaxes(L, 0, spectral_dim)
    return L


# ---------------------------------------------------------------------------------------
# TUD integration                                   radiative_transfer.py:340-392
def tud_from_od(X, OD, T, Z, Altitudes=(500,), theta_r=0.0, N_angle=30, returnOD=False):
    """Body of compute_TUD after the per-layer OD loop (:340-392), quirks 3-6 reproduced:
    nL is overwritten by the LAST altitude's layer count and reused by the downwelling loop;
    tau uses the Z<=zs mask, L-up uses the first count(mask) layers; theta=0 carries weight 0."""
    X = np.asarray(X, dtype=np.float64)
    OD = np.asarray(OD, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    Z = np.asarray(Z, dtype=np.float64)
    nL = T.size
    nA = int(N_angle)
    f = lambda x: np.array([x]).ravel()
    Z_s = f(Altitudes)
    mu_s = f(1.0 / np.cos(theta_r))
    Lu_ = np.zeros((X.size, Z_s.size, mu_s.size))
    Ld_ = np.zeros((X.size, nA))
    tau_ = Lu_.copy()
    B = planckian(X, T)
    for ii, zs in enumerate(Z_s):
        for jj, mu in enumerate(mu_s):
            ix = Z <= zs
            if returnOD:
                tau_[:, ii, jj] = np.sum(OD[:, ix] * mu, axis=1)
            else:
                tau_[:, ii, jj] = np.exp(-1.0 * np.sum(OD[:, ix] * mu, axis=1))
            nL = int(np.sum(ix))
            for kk in range(nL):
                t = np.exp(-OD[:, kk] * mu)
                Lu_[:, ii, jj] = t * Lu_[:, ii, jj] + (1 - t) * B[:, kk]
    if (len(Z_s) == 1) and (len(mu_s) == 1):
        tau_ = tau_[:, 0, 0]
        Lu_ = Lu_[:, 0, 0]
    if (len(Z_s) == 1) and not (len(mu_s) == 1):
        tau_ = tau_[:, 0, :]
        Lu_ = Lu_[:, 0, :]
    if not (len(Z_s) == 1) and (len(mu_s) == 1):
        tau_ = tau_[:, :, 0]
        Lu_ = Lu_[:, :, 0]
    angles = np.linspace(0, np.pi / 2.0, nA, endpoint=False)
    for ii, th in enumerate(angles):
        for jj in np.arange(nL)[::-1]:
            t = np.exp(-OD[:, jj] / np.cos(th))
            Ld_[:, ii] = t * Ld_[:, ii] + (1 - t) * B[:, jj]
    cos_dOmega = np.cos(angles) * np.sin(angles)
    Ld_ = np.sum(Ld_ * cos_dOmega, axis=1) / np.sum(cos_dOmega)
    return tau_, Lu_, Ld_.flatten()


def compute_LWIR_apparent_radiance(X, emis, Ts, tau, La, Ld, dT=None, return_Ls=False):
    """radiative_transfer.py:1017-1069. Dtype-agnostic like the reference (callers feed fp32)."""
    if dT is not None:
        T_ = Ts.flatten()[:, np.newaxis] + np.asarray(dT).flatten()[np.newaxis, :]
        B_ = planckian(X, T_)[:, np.newaxis, :, :]
        tau_ = tau[:, np.newaxis, :, np.newaxis]
        La_ = La[:, np.newaxis, :, np.newaxis]
        Ld_ = Ld[:, np.newaxis, :, np.newaxis]
        em_ = emis[:, :, np.newaxis, np.newaxis]
    else:
        T_ = Ts.flatten()
        B_ = planckian(X, T_)[:, np.newaxis, :]
        tau_ = tau[:, np.newaxis, :]
        La_ = La[:, np.newaxis, :]
        Ld_ = Ld[:, np.newaxis, :]
        em_ = emis[:, :, np.newaxis]
    if return_Ls:
        Ls = em_ * B_ + (1 - em_) * Ld_
        L = tau_ * Ls + La_
        return L, Ls
    return tau_ * (em_ * B_ + (1 - em_) * Ld_) + La_


# ---------------------------------------------------------------------------------------
# MAKO instrument line shape         radiative_transfer.py:1072-1263, ILS_MAKO.py:2-35
# 128 MAKO band centres [um] (instrument constants; :1092-1223)
MAKO_UM = np.array([
    7.5711, 7.6158, 7.6606, 7.7053, 7.7500, 7.7947, 7.8394, 7.8841, 7.9288, 7.9734, 8.0181, 8.0627,
    8.1073, 8.1519, 8.1965, 8.2411, 8.2857, 8.3303, 8.3748, 8.4194, 8.4639, 8.5084, 8.5529, 8.5974,
    8.6419, 8.6863, 8.7308, 8.7752, 8.8197, 8.8641, 8.9085, 8.9529, 8.9973, 9.0417, 9.0860, 9.1304,
    9.1747, 9.2190, 9.2633, 9.3076, 9.3519, 9.3962, 9.4405, 9.4847, 9.5290, 9.5732, 9.6174, 9.6616,
    9.7058, 9.7500, 9.7942, 9.8383, 9.8825, 9.9266, 9.9707, 10.0148, 10.0589, 10.1030, 10.1471,
    10.1912, 10.2352, 10.2792, 10.3233, 10.3673, 10.4113, 10.4553, 10.4993, 10.5432, 10.5872,
    10.6311, 10.6751, 10.7190, 10.7629, 10.8068, 10.8507, 10.8945, 10.9384, 10.9822, 11.0261,
    11.0699, 11.1137, 11.1575, 11.2013, 11.2451, 11.2888, 11.3326, 11.3763, 11.4201, 11.4638,
    11.5075, 11.5512, 11.5948, 11.6385, 11.6822, 11.7258, 11.7694, 11.8131, 11.8567, 11.9003,
    11.9439, 11.9874, 12.0310, 12.0745, 12.1181, 12.1616, 12.2051, 12.2486, 12.2921, 12.3356,
    12.3791, 12.4225, 12.4660, 12.5094, 12.5528, 12.5962, 12.6396, 12.6830, 12.7264, 12.7697,
    12.8131, 12.8564, 12.8997, 12.9430, 12.9863, 13.0296, 13.0729, 13.1162, 13.1594])


def mako_band_axis(X, resFactor=None, clip=True):
    """Band centres in cm^-1: :1226-1233 (triangle variant clips to the open (X.min, X.max))."""
    X_out = MAKO_UM.copy()
    if resFactor is not None:
        _x0 = np.linspace(0, 1, len(X_out))
        _x1 = np.linspace(0, 1, int(len(X_out) * resFactor))
        X_out = np.interp(_x1, _x0, X_out)
    X_out = np.sort(10000.0 / X_out)
    if clip:
        X_out = X_out[(X_out > X.min()) & (X_out < X.max())]
    return X_out


def ILS_MAKO(X, Y, resFactor=None, returnX=True, fwhm_sf=1.0, shift=0.0, scale=1.0):
    """Triangle ILS, radiative_transfer.py:1072-1263. Banded evaluation (the reference's dense
    (nS,nX,nB) temporary is avoided; the sums are the same sums over the non-zero support)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y)
    X_out = mako_band_axis(X, resFactor, clip=True)
    sigma_out = fwhm_sf * np.abs(np.gradient(X_out)) * 1.6
    ctr = scale * X_out + shift
    Y2 = Y[:, None] if Y.ndim == 1 else Y
    Y_out = np.zeros((X_out.size, Y2.shape[1]))
    for b in range(X_out.size):
        # X ascending: support is the open interval |x-c| < s
        lo = np.searchsorted(X, ctr[b] - sigma_out[b], side="right")
        hi = np.searchsorted(X, ctr[b] + sigma_out[b], side="left")
        w = 1.0 - np.abs(X[lo:hi] - ctr[b]) / sigma_out[b]
        w[w < 0] = 0
        with np.errstate(all="ignore"):
            Y_out[b] = (w[:, None] * Y2[lo:hi]).sum(axis=0) / w.sum()
    if Y.ndim == 1:
        Y_out = Y_out[:, 0]
    if returnX:
        return X_out, Y_out
    return Y_out


def ILS_MAKO_gauss(X, Y):
    """Gaussian ILS, ILS_MAKO.py:2-35: sigma=|grad X_out| (no 1.6), no clipping, dense weights."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y)
    X_out = mako_band_axis(X, None, clip=False)
    sigma_out = np.abs(np.gradient(X_out))
    Y2 = Y[:, None] if Y.ndim == 1 else Y
    Y_out = np.zeros((X_out.size, Y2.shape[1]))
    for b in range(X_out.size):
        g = np.exp(-0.5 * ((X - X_out[b]) / sigma_out[b]) ** 2) / (sigma_out[b] * np.sqrt(2.0 * np.pi))
        Y_out[b] = (g[:, None] * Y2).sum(axis=0) / g.sum()
    if Y.ndim == 1:
        Y_out = Y_out[:, 0]
    return X_out, Y_out


# ---------------------------------------------------------------------------------------
# hapi line-by-line chain
def AtoB(aa, A, B, npt):
    """Lagrange 3-/4-point interpolation, misc/hapi.py:5311-5388 (TIPS-2011 scheme)."""
    for I in range(2, npt + 1):
        if A[I - 1] >= aa:
            if I < 3 or I == npt:
                J = 3 if I < 3 else npt
                J -= 1
                a0, a1, a2 = A[J - 2], A[J - 1], A[J]
                A0 = (aa - a1) * (aa - a2) / ((a0 - a1) * (a0 - a2))
                A1 = (aa - a0) * (aa - a2) / ((a1 - a0) * (a1 - a2))
                A2 = (aa - a0) * (aa - a1) / ((a2 - a0) * (a2 - a1))
                return A0 * B[J - 2] + A1 * B[J - 1] + A2 * B[J]
            J = I - 1
            a0, a1, a2, a3 = A[J - 2], A[J - 1], A[J], A[J + 1]
            A0 = (aa - a1) * (aa - a2) * (aa - a3)
            A0 = A0 / ((a0 - a1) * (a0 - a2) * (a0 - a3))
            A1 = (aa - a0) * (aa - a2) * (aa - a3)
            A1 = A1 / ((a1 - a0) * (a1 - a2) * (a1 - a3))
            A2 = (aa - a0) * (aa - a1) * (aa - a3)
            A2 = A2 / ((a2 - a0) * (a2 - a1) * (a2 - a3))
            A3 = (aa - a0) * (aa - a1) * (aa - a2)
            A3 = A3 / ((a3 - a0) * (a3 - a1) * (a3 - a2))
            return A0 * B[J - 2] + A1 * B[J - 1] + A2 * B[J] + A3 * B[J + 1]
    raise ValueError("AtoB: aa above the last node")


def PYTIPS(M, I, T):
    """BD_TIPS_2011_PYTHON(M,I,T)[1], misc/hapi.py:9568-9582,10030."""
    if T < 70.0 or T > 3000.0:
        raise Exception("TIPS: T must be between 70K and 3000K.")
    t = _tables()
    try:
        q = t["q"][(int(M), int(I))]
    except KeyError:
        raise Exception("TIPS: no data for M,I = %d,%d." % (M, I))
    return AtoB(T, t["tdat"], q, t["tdat"].size)


def EnvironmentDependency_Intensity(S, T, Tref, SigmaT, SigmaTref, Elower, nu):
    """misc/hapi.py:10169-10175."""
    ch = np.exp(-HAPI_C2 * Elower / T) * (1 - np.exp(-HAPI_C2 * nu / T))
    zn = np.exp(-HAPI_C2 * Elower / Tref) * (1 - np.exp(-HAPI_C2 * nu / Tref))
    return S * SigmaTref / SigmaT * ch / zn


def volumeConcentration(p, T):
    """misc/hapi.py:10163-10164 (CGS, molecules/cm^3)."""
    return (p / 9.869233e-7) / (CBOLTS * T)


def weideman_coeffs(N=24):
    """Coefficients of Weideman's rational expansion (SIAM J. Numer. Anal. 31, 1994), as rebuilt
    per call by misc/hapi.py:9812-9824. polyval order (highest power first)."""
    M = 2 * N
    M2 = 2 * M
    k = np.arange(-M + 1, M)
    L = np.sqrt(N / np.sqrt(2))
    theta = k * np.pi / M
    t = L * np.tan(theta / 2)
    f = np.zeros(len(t) + 1)
    f[1:] = np.exp(-t ** 2) * (L ** 2 + t ** 2)
    a = np.real(np.fft.fft(np.fft.fftshift(f))) / M2
    return np.flipud(a[1:N + 1]), L


_W24, _L24 = weideman_coeffs(24)


def hum1_wei(x, y):
    """misc/hapi.py:9833-9844: Weideman-24 where |x|+y<15, else the 1-term asymptote."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64) + np.zeros_like(x)
    t = y - 1.0j * x
    cerf = 1 / np.sqrt(np.pi) * t / (0.5 + t ** 2)
    mask = abs(x) + y < 15.0
    if np.any(mask):
        z = x[mask] + 1.0j * y[mask]
        Z = (_L24 + 1.0j * z) / (_L24 - 1.0j * z)
        p = np.polyval(_W24, Z)
        w = 2 * p / (_L24 - 1.0j * z) ** 2 + (1 / np.sqrt(np.pi)) / (_L24 - 1.0j * z)
        np.place(cerf, mask, w)
    return cerf.real, cerf.imag


def PROFILE_VOIGT(sg0, GamD, Gam0, sg):
    """misc/hapi.py:10131-10140 -> pcqsdhc PART1 (:9900-9915) + common part (:10022):
    LS = (1/pi) * sqrt(pi)*cte*w(x+iy),  x=(sg-sg0)*cte, y=Gam0*cte, cte=sqrt(ln2)/GamD."""
    sg = np.asarray(sg, dtype=np.float64)
    cte = np.sqrt(np.log(2.0)) / GamD
    rpi = np.sqrt(np.pi)
    Z1 = (1.0j * (sg0 - sg) + complex(Gam0)) * cte
    WR1, WI1 = hum1_wei(-Z1.imag, Z1.real)
    A = rpi * cte * (WR1 + 1.0j * WI1)
    LS = (1.0 / np.pi) * (A / (1.0 - 0.0 * A))
    return LS.real, LS.imag


def line_params(tbl, T, p, Diluent=None, GammaL="gamma_air"):
    """Per-line environment dependences, misc/hapi.py:11068-11131 (vectorised over lines).
    Returns dict of arrays: S(T) (without abundance factors), GammaD, Gamma0, Shift0."""
    t = _tables()
    nu = np.asarray(tbl["nu"], dtype=np.float64)
    M = np.asarray(tbl["molec_id"]).astype(int)
    I = np.asarray(tbl["local_iso_id"]).astype(int)
    n = nu.size
    qT = {}
    qR = {}
    mass = np.zeros(n)
    SigmaT = np.zeros(n)
    SigmaR = np.zeros(n)
    for mi in sorted(set(zip(M.tolist(), I.tolist()))):
        qT[mi] = PYTIPS(mi[0], mi[1], T)
        qR[mi] = PYTIPS(mi[0], mi[1], TREF)
        sel = (M == mi[0]) & (I == mi[1])
        SigmaT[sel] = qT[mi]
        SigmaR[sel] = qR[mi]
        mass[sel] = t["mass"][mi]
    S = EnvironmentDependency_Intensity(np.asarray(tbl["sw"], dtype=np.float64), T, TREF, SigmaT, SigmaR,
                                        np.asarray(tbl["elower"], dtype=np.float64), nu)
    m = mass * CMASSMOL * 1000
    GammaD = np.sqrt(2 * CBOLTS * T * np.log(2) / m / CC ** 2) * nu
    if not Diluent:
        Diluent = {"air": 1.0} if GammaL.lower() == "gamma_air" else {"self": 1.0}
    Gamma0 = np.zeros(n)
    Shift0 = np.zeros(n)
    for species, abun in Diluent.items():
        sp = species.lower()
        g = np.asarray(tbl.get("gamma_" + sp, np.zeros(n)), dtype=np.float64)
        if "n_" + sp in tbl:
            npow = np.asarray(tbl["n_" + sp], dtype=np.float64).copy()
            if sp == "self":
                z = npow == 0.0
                npow[z] = np.asarray(tbl["n_air"], dtype=np.float64)[z]
        else:
            npow = np.asarray(tbl["n_air"], dtype=np.float64)
        Gamma0 = Gamma0 + abun * (g * p / PREF * (TREF / T) ** npow)
        d = np.asarray(tbl.get("delta_" + sp, np.zeros(n)), dtype=np.float64)
        dp = np.asarray(tbl.get("deltap_" + sp, np.zeros(n)), dtype=np.float64)
        Shift0 = Shift0 + abun * ((d + dp * (T - TREF)) * p / PREF)
    return {"S": S, "GammaD": GammaD, "Gamma0": Gamma0, "Shift0": Shift0, "M": M, "I": I}


def PROFILE_LORENTZ(sg0, Gam0, sg):
    """misc/hapi.py:10142-10150."""
    return Gam0 / (np.pi * (Gam0 ** 2 + (sg - sg0) ** 2))


def PROFILE_DOPPLER(sg0, GamD, sg):
    """misc/hapi.py:10152-10160 (cSqrtLn2divSqrtPi = sqrt(ln2/pi), cLn2 = ln2)."""
    return 0.469718639319144059835 * np.exp(-0.6931471805599 * ((sg - sg0) / GamD) ** 2) / GamD


def absorptionCoefficient_Lorentz(tbl, **kw):
    """misc/hapi.py:11144-11375: the Voigt loop with PROFILE_LORENTZ and OmegaWingF = max(OmegaWing, HW*Gamma0)."""
    return absorptionCoefficient_Voigt(tbl, _profile="lorentz", **kw)


def absorptionCoefficient_Doppler(tbl, LineShift=True, **kw):
    """misc/hapi.py:11384-11559: PROFILE_DOPPLER, GammaD from the function's own SI constants (:11534-11538),
    OmegaWingF = max(OmegaWing, HW*GammaD), Shift0 = delta_air*p/pref when LineShift (:11510-11513, 11543)."""
    return absorptionCoefficient_Voigt(tbl, _profile="doppler", _lineshift=LineShift, **kw)


def absorptionCoefficient_Voigt(tbl, Components=None, T=296.0, p=1.0, OmegaGrid=None, OmegaWing=0.0,
                                OmegaWingHW=50.0, HITRAN_units=True, GammaL="gamma_air", Diluent=None,
                                IntensityThreshold=0.0, _profile="voigt", _lineshift=True):
    """misc/hapi.py:10906-11141 on an explicit grid. `tbl` is the column dict of the line table
    (LOCAL_TABLE_CACHE[name]['data'], :438-463). Components: list of (M,I[,abundance]); None = every
    (M,I) in the table at natural abundance (:10237-10251). Returns (Omegas, Xsect)."""
    t = _tables()
    Omegas = np.sort(np.asarray(OmegaGrid, dtype=np.float64))
    Xsect = np.zeros(Omegas.size)
    Mcol = np.asarray(tbl["molec_id"]).astype(int)
    Icol = np.asarray(tbl["local_iso_id"]).astype(int)
    if Components is None:
        Components = sorted(set(zip(Mcol.tolist(), Icol.tolist())))
    ABUN, NAT = {}, {}
    for c in Components:
        mi = (int(c[0]), int(c[1]))
        if mi not in t["abun"]:
            raise Exception("cannot find component M,I = %d,%d." % mi)
        ABUN[mi] = c[2] if len(c) >= 3 else t["abun"][mi]
        NAT[mi] = t["abun"][mi]
    factor = 1.0 if HITRAN_units else volumeConcentration(p, T)
    keep = np.array([(m, i) in ABUN for m, i in zip(Mcol.tolist(), Icol.tolist())], dtype=bool)
    sub = {k: np.asarray(v)[keep] for k, v in tbl.items()}
    if sub["nu"].size == 0:
        return Omegas, Xsect
    P = line_params(sub, T, p, Diluent=Diluent, GammaL=GammaL)
    nu = np.asarray(sub["nu"], dtype=np.float64)
    glist = Omegas.tolist()
    for r in range(nu.size):
        S = P["S"][r]
        if S < IntensityThreshold:
            continue
        GammaD, Gamma0, Shift0 = P["GammaD"][r], P["Gamma0"][r], P["Shift0"][r]
        if _profile == "doppler":
            mass = t["mass"][(int(P["M"][r]), int(P["I"][r]))]
            GammaD = (1.1774100225 / 2.99792458e8) * np.sqrt(1.3806503e-23 / 1.66053873e-27) * np.sqrt(T) * nu[r] / np.sqrt(mass)
            Shift0 = (float(sub["delta_air"][r]) if _lineshift else 0.0) * p / PREF
            W = max(OmegaWing, OmegaWingHW * GammaD)
        elif _profile == "lorentz":
            W = max(OmegaWing, OmegaWingHW * Gamma0)
        else:
            W = max(OmegaWing, OmegaWingHW * Gamma0, OmegaWingHW * GammaD)
        lo = _bisect.bisect(glist, nu[r] - W)
        hi = _bisect.bisect(glist, nu[r] + W)
        if hi <= lo:
            continue
        if _profile == "sdvoigt":
            dil = Diluent if Diluent else ({"air": 1.0} if GammaL.lower() == "gamma_air" else {"self": 1.0})
            Gamma2 = 0.0
            for species, abun in dil.items():
                sp = species.lower()
                sd = float(sub["SD_" + sp][r]) if ("SD_" + sp) in sub else 0.0
                g0db = float(sub["gamma_" + sp][r]) if ("gamma_" + sp) in sub else 0.0
                Gamma2 += abun * (sd * p / PREF) * g0db
            ls = PROFILE_SDVOIGT(nu[r], GammaD, Gamma0, Gamma2, Shift0, 0.0, Omegas[lo:hi])
        elif _profile == "doppler":
            ls = PROFILE_DOPPLER(nu[r] + Shift0, GammaD, Omegas[lo:hi])
        elif _profile == "lorentz":
            ls = PROFILE_LORENTZ(nu[r] + Shift0, Gamma0, Omegas[lo:hi])
        else:
            ls = PROFILE_VOIGT(nu[r] + Shift0, GammaD, Gamma0, Omegas[lo:hi])[0]
        mi = (int(P["M"][r]), int(P["I"][r]))
        Xsect[lo:hi] += factor / NAT[mi] * ABUN[mi] * S * ls
    return Omegas, Xsect


def layer_od(tbl, X, T, P_pa, PL_km, MF_VAL, MF_ID):
    """Optical depth of one homogeneous layer -- the build-defined meaning of compute_OD
    (radiative_transfer.py:395-456 cannot run: LBLRTM is an LFS stub):
    OD = sum_m  k_m(nu; T, p) [cm^-1, HITRAN_units=False] * x_m * PL * 1e5 cm."""
    X = np.asarray(X, dtype=np.float64)
    od = np.zeros(X.size)
    Mcol = np.asarray(tbl["molec_id"]).astype(int)
    Icol = np.asarray(tbl["local_iso_id"]).astype(int)
    pairs = sorted(set(zip(Mcol.tolist(), Icol.tolist())))
    for m, ppmv in zip(np.asarray(MF_ID).tolist(), np.asarray(MF_VAL).tolist()):
        comps = [(mm, ii) for (mm, ii) in pairs if mm == m]
        if not comps:
            continue
        _, xs = absorptionCoefficient_Voigt(tbl, Components=comps, T=float(T), p=float(P_pa) / 101325.0,
                                            OmegaGrid=X, HITRAN_units=False)
        od += xs * (ppmv * 1e-6) * PL_km * 1e5
    return od


def compute_TUD(tbl, Xmin, Xmax, DVOUT, Zs, Ts, Ps, PLs, MFs_VAL, MFs_ID, Altitudes=(500,), theta_r=0.0,
                N_angle=30, returnOD=False, return_layers=False):
    """radiative_transfer.py:274-392 with compute_OD := layer_od (see above)."""
    X = make_spectral_axis(Xmin, Xmax, DVOUT)
    nL = np.asarray(Ts).size
    OD = np.zeros((X.size, nL))
    for ii in range(nL):
        OD[:, ii] = layer_od(tbl, X, Ts[ii], Ps[ii], PLs[ii], np.asarray(MFs_VAL)[ii, :], MFs_ID)
    tau, Lu, Ld = tud_from_od(X, OD, Ts, Zs, Altitudes, theta_r, N_angle, returnOD)
    if return_layers:
        return X, tau, Lu, Ld, OD
    return X, tau, Lu, Ld


# ---- post-processing of TUD products ----------------------------------------
def smooth(x, window_len=11, window="hanning"):
    """radiative_transfer.py:1266-1324: reflect-pad by window_len-1 samples, convolve with the normalised
    window ('valid'), cut back to len(x). For an even window_len the result is half a sample off centre."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 1 or x.size < window_len or window_len < 3:
        return x
    if window not in ("flat", "hanning", "hamming", "bartlett", "blackman"):
        return x
    s = np.r_[x[window_len - 1:0:-1], x, x[-2:-window_len - 1:-1]]  # :1314
    w = np.ones(window_len, "d") if window == "flat" else getattr(np, window)(window_len)
    y = np.convolve(w / w.sum(), s, mode="valid")
    ix0 = int(np.ceil(window_len / 2 - 1))
    ix1 = -int(np.floor(window_len / 2))
    return y[ix0:ix1]


def smooth_sym(y, window_len, window="hanning"):
    """The symmetrised smoother of reduceResolution (:1331): mean of smoothing y and smoothing y reversed."""
    return 0.5 * (smooth(y, window_len, window) + smooth(y[::-1], window_len, window)[::-1])


def reduceResolution(X, Y, dX, N=4, window="hanning", X_out=None):
    """radiative_transfer.py:1327-1350 (np.int -> int): symmetric window smoothing over round(dX/dX_in) samples,
    then scipy's cubic interp1d (a not-a-knot cubic spline through ALL smoothed samples) evaluated on
    X_out = linspace(X_[smFactor], X_[-smFactor-1], ceil(N*span/dX)+1)."""
    import scipy.interpolate

    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    dX_in = np.mean(np.diff(X))
    smFactor = int(np.round(dX / dX_in))
    sm = lambda y: smooth_sym(y, smFactor, window)
    interp = lambda x, y, x0: scipy.interpolate.interp1d(x, y, kind="cubic", bounds_error=False, fill_value="extrapolate")(x0)
    X_ = sm(X)
    nPts = int(np.ceil(N * (X_[-smFactor - 1] - X_[smFactor]) / dX)) + 1
    ret_x = X_out is None
    if ret_x:
        X_out = np.linspace(X_[smFactor], X_[-smFactor - 1], nPts)
    if Y.ndim > 1:
        Y_out = np.zeros((X_out.size, Y.shape[-1]))
        for ii in range(Y.shape[-1]):
            Y_out[:, ii] = interp(X_, sm(Y[:, ii]), X_out)
    else:
        Y_out = interp(X_, sm(Y), X_out)
    return (X_out, Y_out) if ret_x else Y_out


# ---- speed-dependent Voigt: pcqsdhc with anuVC = eta = 0 -------------------------------
_CPF3_TT = np.array([0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5, 9.5, 10.5, 11.5, 12.5, 13.5, 14.5])


def cpf3(X, Y):
    """misc/hapi.py:9645-9670: 15-term asymptotic series of w(z), z = X + iY."""
    zm1 = 1.0 / (np.asarray(X, dtype=np.float64) + 1.0j * np.asarray(Y, dtype=np.float64))
    zm2 = zm1 ** 2
    zsum = np.ones_like(zm1)
    zterm = np.ones_like(zm1)
    for t in _CPF3_TT:
        zterm = zterm * zm2 * t
        zsum = zsum + zterm
    zsum = zsum * 1.0j * zm1 * 0.564189583547756
    return zsum.real, zsum.imag


def PROFILE_SDVOIGT(sg0, GamD, Gam0, Gam2, Shift0, Shift2, sg):
    """misc/hapi.py:10117-10129 -> pcqsdhc (:9850-10024) with anuVC = eta = 0, for which the common part (:10022) is
    LS = Aterm/pi. Real part only (what absorptionCoefficient_SDVoigt uses, :10897). CPF = hum1_wei (:9846)."""
    sg = np.asarray(sg, dtype=np.float64)
    cte = np.sqrt(np.log(2.0)) / GamD
    rpi = np.sqrt(np.pi)
    c0 = complex(Gam0, Shift0)
    c2 = complex(Gam2, Shift2)
    c0t = c0 - 1.5 * c2
    c2t = c2
    cw = lambda x, y: (lambda r: r[0] + 1.0j * r[1])(hum1_wei(x, y))
    if abs(c2t) == 0.0:  # PART1 (:9908-9915)
        Z1 = (1.0j * (sg0 - sg) + c0t) * cte
        A = rpi * cte * cw(-Z1.imag, Z1.real)
        return (A / np.pi).real
    X = (1.0j * (sg0 - sg) + c0t) / c2t
    Y = 1.0 / ((2.0 * cte * c2t)) ** 2
    csqrtY = (Gam2 - 1.0j * Shift2) / (2.0 * cte * (Gam2 ** 2 + Shift2 ** 2))
    p2 = np.abs(X) <= 3.0e-8 * abs(Y)
    p3 = (abs(Y) <= 1.0e-15 * np.abs(X)) & ~p2
    p4 = ~(p2 | p3)
    A = np.zeros(sg.size, dtype=np.complex128)
    if np.any(p4):  # PART4 (:9933-9971)
        Z1 = np.sqrt(X[p4] + Y) - csqrtY
        Z2 = Z1 + 2.0 * csqrtY
        x1, y1, x2, y2 = -Z1.imag, Z1.real, -Z2.imag, Z2.real
        S1, S2 = np.sqrt(x1 ** 2 + y1 ** 2), np.sqrt(x2 ** 2 + y2 ** 2)
        use3 = (np.abs(S1 - S2) <= 1.0) & (np.maximum(S1, S2) > 8.0) & (np.minimum(S1, S2) <= 8.0)
        W1 = np.zeros(Z1.size, dtype=np.complex128)
        W2 = np.zeros(Z1.size, dtype=np.complex128)
        if np.any(use3):
            r1, r2 = cpf3(x1[use3], y1[use3]), cpf3(x2[use3], y2[use3])
            W1[use3], W2[use3] = r1[0] + 1.0j * r1[1], r2[0] + 1.0j * r2[1]
        if np.any(~use3):
            W1[~use3], W2[~use3] = cw(x1[~use3], y1[~use3]), cw(x2[~use3], y2[~use3])
        A[p4] = rpi * cte * (W1 - W2)
    if np.any(p2):  # PART2 (:9974-9989)
        Z1 = (1.0j * (sg0 - sg[p2]) + c0t) * cte
        Z2 = np.sqrt(X[p2] + Y) + csqrtY
        A[p2] = rpi * cte * (cw(-Z1.imag, Z1.real) - cw(-Z2.imag, Z2.real))
    if np.any(p3):  # PART3 (:9992-10017; the reference indexes the full X at :10003, which only works when every point is PART3)
        Xt = X[p3]
        sq = np.sqrt(Xt)
        near = np.abs(sq) <= 4.0e3
        At = np.zeros(Xt.size, dtype=np.complex128)
        if np.any(near):
            At[near] = (2.0 * rpi / c2t) * (1.0 / rpi - sq[near] * cw(-sq[near].imag, sq[near].real))
        if np.any(~near):
            At[~near] = (1.0 / c2t) * (1.0 / Xt[~near] - 1.5 / (Xt[~near] ** 2))
        A[p3] = At
    return (A / np.pi).real


def absorptionCoefficient_SDVoigt(tbl, **kw):
    """misc/hapi.py:10657-10904: the Voigt loop with PROFILE_SDVOIGT(nu, GammaD, Gamma0, Gamma2, Shift0, 0, grid) and
    Gamma2 = sum_species abun * SD_species * p/pref * gamma_species(296 K) (:10884-10890)."""
    return absorptionCoefficient_Voigt(tbl, _profile="sdvoigt", **kw)
