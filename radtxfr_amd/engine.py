"""Host-side engine: PyTorch-ROCm owns device memory and streams, libradtxfr_hip.so does the work.

This is plumbing between the reference-shaped shims (radiative_transfer.py, hapi.py, ILS_MAKO.py in
this package) and the C ABI (include/radtxfr_hip.h). Every compute call goes through the HIP
library; nothing here falls back to NumPy or to the oracle.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, tips

TUD_MAX_MU = 8  # slant paths per rtx_tud launch (include/radtxfr_hip.h)

# hapi constants used for per-species / per-layer host factors (misc/hapi.py:84-92, 10163-10164)
CBOLTS = 1.380648813e-16
TREF = 296.0


def volumeConcentration(p, T):
    """Molecules/cm^3 at p [atm], T [K] (misc/hapi.py:10163-10164)."""
    return (p / 9.869233e-7) / (CBOLTS * T)


class trace_range:
    """roctx range around a stage (tracing hook), visible in `rocprofv3 --marker-trace`: enabled with
    RADTXFR_ROCTX=1, otherwise a no-op that costs one attribute test. torch.cuda.nvtx maps to roctx on ROCm."""
    enabled = bool(int(__import__("os").environ.get("RADTXFR_ROCTX", "0") or 0))

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if trace_range.enabled:
            torch.cuda.nvtx.range_push(self.name)

    def __exit__(self, *exc):
        if trace_range.enabled:
            torch.cuda.nvtx.range_pop()
        return False


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.RtxError("no HIP device visible: radtxfr_amd has no CPU fallback (torch.cuda.is_available() is False)")


def device(dev=None):
    require_gpu()
    if dev is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(dev)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _h(a):
    """contiguous float64 host array + its pointer (kept alive by the caller holding the array)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.c_void_p)


class Grid:
    """np.linspace(xmin, xmax, n_total) (radiative_transfer.py:269-270), optionally one shard of it."""

    def __init__(self, xmin, xmax, n_total, offset=0, n=None):
        self.c = _lib.make_grid(xmin, xmax, n_total, offset, n)

    xmin = property(lambda s: s.c.xmin)
    xmax = property(lambda s: s.c.xmax)
    step = property(lambda s: s.c.step)
    n_total = property(lambda s: s.c.n_total)
    offset = property(lambda s: s.c.offset)
    n = property(lambda s: s.c.n)

    def shard(self, offset, n):
        return Grid(self.xmin, self.xmax, self.n_total, offset, n)

    def axis(self):
        """Materialise this shard's wavenumbers (host fp64), bit-identical to np.linspace."""
        ig = np.arange(self.offset, self.offset + self.n, dtype=np.float64)
        X = ig * self.step + self.xmin  # arange*step + start: the two roundings np.linspace makes
        if self.n and self.offset + self.n == self.n_total:
            X[-1] = self.xmax
        return X

    def x_at(self, i):
        """Wavenumber of local index i (same bits as axis()[i])."""
        ig = self.offset + int(i)
        return self.xmax if ig == self.n_total - 1 else float(np.float64(ig) * self.step + self.xmin)

    @staticmethod
    def from_axis(X, rtol=1e-9):
        """Recognise a uniform ascending axis; raises if X is not np.linspace-like."""
        X = np.asarray(X, dtype=np.float64).ravel()
        if X.size < 2:
            raise ValueError("spectral axis needs at least 2 points")
        g = Grid(X[0], X[-1], X.size)
        dev = np.max(np.abs(X - np.linspace(X[0], X[-1], X.size)))
        if not (g.step > 0) or dev > rtol * abs(g.step):
            raise NotImplementedError("the HIP line-sum needs a uniform ascending wavenumber grid "
                                      f"(max deviation from np.linspace = {dev:g}, step = {g.step:g})")
        return g

    def byref(self):
        return C.byref(self.c)


_COLS = ("nu", "sw", "elower", "gamma_air", "gamma_self", "n_air", "delta_air")
_OPT_COLS = ("n_self", "deltap_air", "delta_self")


class LineTable:
    """Device-resident HITRAN-format line table, sorted by nu (include/radtxfr_hip.h: rtx_lines).
    `columns` is the column dict of the reference's table type: LOCAL_TABLE_CACHE[name]['data']
    (misc/hapi.py:438-463)."""

    def __init__(self, columns):
        require_gpu()
        lib = _lib.load()
        self.device = torch.cuda.current_device()  # the table lives on the device that is current at creation
        nu = np.asarray(columns["nu"], dtype=np.float64)
        order = np.argsort(nu, kind="stable")
        self.n = int(nu.size)
        self.cols = {k: np.ascontiguousarray(np.asarray(columns[k], dtype=np.float64)[order]) for k in _COLS}
        for k in _OPT_COLS:
            if k in columns:
                self.cols[k] = np.ascontiguousarray(np.asarray(columns[k], dtype=np.float64)[order])
        M = np.asarray(columns["molec_id"]).astype(np.int64)[order]
        I = np.asarray(columns["local_iso_id"]).astype(np.int64)[order]
        self.molec_id, self.local_iso_id = M, I
        # distinct (molec_id, local_iso_id) pairs in sorted order and each row's index into them, vectorised: 100 000 Python
        # tuples per table were enough to tip the interpreter into a full garbage collection (40 ms with torch loaded)
        code = M * 1000003 + I  # ascending code = ascending (molec_id, local_iso_id) for the non-negative ids HITRAN uses
        uniq, inv = np.unique(code, return_inverse=True)
        pairs = [(int(u // 1000003), int(u % 1000003)) for u in uniq]
        self.species = pairs if pairs else [(0, 0)]
        sp = np.ascontiguousarray(inv, dtype=np.int32)
        self._h = C.c_void_p(0)
        ptr = lambda k: self.cols[k].ctypes.data_as(C.c_void_p) if k in self.cols else C.c_void_p(0)
        _lib.check(lib.rtx_lines_create(
            self.n, len(self.species), ptr("nu"), ptr("sw"), ptr("elower"), ptr("gamma_air"), ptr("gamma_self"),
            ptr("n_air"), ptr("n_self"), ptr("delta_air"), ptr("deltap_air"), ptr("delta_self"),
            sp.ctypes.data_as(C.c_void_p), C.byref(self._h)))
        # optional speed-dependence columns (absorptionCoefficient_SDVoigt, misc/hapi.py:10884-10887)
        self.has_sd = False
        sd = {}
        for k in ("SD_air", "SD_self"):
            if k in columns:
                sd[k] = np.ascontiguousarray(np.asarray(columns[k], dtype=np.float64)[order])
                self.has_sd = self.has_sd or bool(np.any(sd[k] != 0.0))
        self._sd = sd
        if self.has_sd:
            sp_ = lambda k: sd[k].ctypes.data_as(C.c_void_p) if k in sd else C.c_void_p(0)
            _lib.check(lib.rtx_lines_set_sd(self._h, sp_("SD_air"), sp_("SD_self")))
        if "deltap_self" in columns:  # misc/hapi.py:11120-11124
            dps = np.ascontiguousarray(np.asarray(columns["deltap_self"], dtype=np.float64)[order])
            if np.any(dps != 0.0):
                self.cols["deltap_self"] = dps
                _lib.check(lib.rtx_lines_set_deltap_self(self._h, dps.ctypes.data_as(C.c_void_p)))
        self._plans = {}

    def host_columns(self):
        """The uploaded columns as a host column dict (sorted by nu): what another device's copy is built from."""
        cols = dict(self.cols)
        cols.update(self._sd)
        cols["molec_id"], cols["local_iso_id"] = self.molec_id, self.local_iso_id
        return cols

    def on_device(self, dev):
        """This table on device index `dev`: itself, or a cached copy uploaded there (closed with this table)."""
        dev = int(dev)
        if dev == self.device:
            return self
        peers = self.__dict__.setdefault("_peers", {})
        if dev not in peers:
            with torch.cuda.device(dev):
                peers[dev] = LineTable(self.host_columns())
        return peers[dev]

    def plan(self, n_layers, n_points):
        """A prep object big enough for (n_layers, n_points); cached, grown on demand."""
        best = None
        for (L, Np), p in self._plans.items():
            if L >= n_layers and Np >= n_points:
                best = p
        if best is None:
            best = VoigtPlan(self, n_layers, n_points)
            self._plans = {k: v for k, v in self._plans.items() if not (k[0] <= n_layers and k[1] <= n_points)}
            self._plans[(n_layers, n_points)] = best
        return best

    def close(self):
        for t in self.__dict__.pop("_peers", {}).values():
            t.close()
        for p in self._plans.values():
            p.close()
        self._plans = {}
        if self._h:
            with torch.cuda.device(self.device):
                _lib.load().rtx_lines_free(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VoigtPlan:
    """Per-(line, layer) record storage (rtx_prep)."""

    def __init__(self, lines, max_layers, max_points):
        self.lines = lines
        self._h = C.c_void_p(0)
        _lib.check(_lib.load().rtx_prep_create(lines._h, int(max_layers), int(max_points), C.byref(self._h)))

    def close(self):
        if self._h:
            _lib.load().rtx_prep_free(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def species_factors(species, T_layers, partitionFunction=None, weight=None):
    """qratio[nS][nL] = Q(Tref)/Q(T_k) (misc/hapi.py:11069-11070) and mass[nS] (:11086).
    A species whose weight[s][:] is all zero is filtered out by the reference BEFORE its partition sums and mass are
    looked up (`continue`, misc/hapi.py:11066): it keeps q = mass = 1 here, so an unselected isotopologue without TIPS
    data, or outside the 70-3000 K range, does not raise."""
    nS, nL = len(species), len(T_layers)
    q = np.ones((nS, nL))
    mass = np.ones(nS)
    if weight is None:
        use = [s for s, mi in enumerate(species) if mi != (0, 0)]
    else:
        live = np.asarray(weight).reshape(nS, -1).any(axis=1)
        use = [s for s, mi in enumerate(species) if mi != (0, 0) and live[s]]
    if not use:
        return q, mass
    if partitionFunction is None or partitionFunction is tips.PYTIPS:
        # default TIPS-2011: all (species, layer) pairs in one vectorised pass, plus Q(Tref) as an extra column
        key = tuple(species[s] for s in use)
        plan = _TIPS_PLANS.get(key)
        if plan is None:
            if len(_TIPS_PLANS) > 32:
                _TIPS_PLANS.clear()
            plan = _TIPS_PLANS[key] = (tips.PartitionPlan(key), np.array([tips.molecularMass(*mi) for mi in key]))
        Tq = np.empty(nL + 1)
        Tq[:nL] = T_layers
        Tq[nL] = TREF
        Q = plan[0](Tq)
        q[use] = Q[:, -1:] / Q[:, :-1]
        mass[use] = plan[1]
    else:
        for s in use:
            m, i = species[s]
            mass[s] = tips.molecularMass(m, i)
            qref = partitionFunction(m, i, TREF)
            for k, T in enumerate(T_layers):
                q[s, k] = qref / partitionFunction(m, i, float(T))
    return q, mass


_TIPS_PLANS = {}


def voigt_sum(lines, grid, T, p_atm, weight, out_f32=None, out_f64=None, dil_air=1.0, dil_self=0.0, omega_wing=0.0,
              omega_wing_hw=50.0, intensity_threshold=0.0, scale=1.0, partitionFunction=None, qratio=None, mass=None,
              profile=0):
    """Prologue + line-sum for n_layers homogeneous states on `grid` (rtx_line_prep_profile + rtx_voigt_sum);
    profile 0 Voigt, 1 Lorentz, 2 Doppler, 3 speed-dependent Voigt (include/radtxfr_hip.h; 3 needs SD columns in `lines`
    and runs rtx_sdvoigt_sum).
    weight[nS][nL] multiplies S(T) per species and layer. Outputs are [nL][grid.n] device tensors."""
    lib = _lib.load()
    T = np.atleast_1d(np.asarray(T, dtype=np.float64))
    p_atm = np.atleast_1d(np.asarray(p_atm, dtype=np.float64))
    nL = T.size
    if qratio is None:
        qratio, mass = species_factors(lines.species, T, partitionFunction, weight=weight)
    plan = lines.plan(nL, grid.n)
    T_h, T_p = _h(T)
    p_h, p_p = _h(p_atm)
    q_h, q_p = _h(qratio)
    w_h, w_p = _h(np.broadcast_to(weight, (len(lines.species), nL)))
    m_h, m_p = _h(mass)
    st = _stream_ptr()
    _lib.check(lib.rtx_line_prep_profile(plan._h, lines._h, grid.byref(), nL, T_p, p_p, q_p, w_p, m_p, float(dil_air),
                                         float(dil_self), float(omega_wing), float(omega_wing_hw),
                                         float(intensity_threshold), float(scale), int(profile), st))
    ld = grid.n
    for o, dt in ((out_f32, torch.float32), (out_f64, torch.float64)):
        if o is not None:
            assert o.dtype == dt and o.is_cuda and o.is_contiguous() and o.shape == (nL, ld), (o.dtype, o.shape)
    if int(profile) == 3:  # speed-dependent Voigt: its own fp64 line-sum
        _lib.check(lib.rtx_sdvoigt_sum(plan._h, grid.byref(), nL, _ptr(out_f32), _ptr(out_f64), ld, st))
    else:
        _lib.check(lib.rtx_voigt_sum(plan._h, grid.byref(), nL, _ptr(out_f32), _ptr(out_f64), ld, st))
    return out_f32, out_f64


_MF_COLUMNS = {}


def layer_weights_od(species, T, P_pa, PL_km, MF_VAL, MF_ID):
    """weight[nS][nL] for optical depth: n(p,T) * x_m * PL*1e5 for the line's molecule.
    `xs * (ppmv*1e-6) * PL * 1e5` with xs carrying factor = volumeConcentration (HITRAN_units=False); the same operations in
    the same order for every species (one [nS][nL] pass: the species -> mixing-ratio column map is cached)."""
    T = np.asarray(T, dtype=np.float64)
    p_atm = np.asarray(P_pa, dtype=np.float64) / 101325.0
    PL = np.asarray(PL_km, dtype=np.float64)
    MF_VAL = np.asarray(MF_VAL, dtype=np.float64).reshape(T.size, -1)
    ids = tuple(int(v) for v in np.asarray(MF_ID).ravel())
    key = (tuple(species), ids)
    m = _MF_COLUMNS.get(key)
    if m is None:
        if len(_MF_COLUMNS) > 64:
            _MF_COLUMNS.clear()
        col = np.array([ids.index(mm) if mm in ids else 0 for mm, _ in species], dtype=np.int64)
        live = np.array([mm in ids for mm, _ in species], dtype=bool)
        m = _MF_COLUMNS[key] = (col, live, bool(live.all()))
    col, live, all_live = m
    nvol = volumeConcentration(p_atm, T)
    w = nvol * (MF_VAL.T[col] * 1e-6) * PL * 1e5 if len(species) else np.zeros((0, T.size))
    if not all_live:
        w[~live] = 0.0
    return w, p_atm


def optical_depths(lines, grid, T, P_pa, PL_km, MF_VAL, MF_ID, out=None):
    """OD[nL][grid.n] float32 on the device (layer-major, wavenumber-contiguous)."""
    T = np.atleast_1d(np.asarray(T, dtype=np.float64))
    w, p_atm = layer_weights_od(lines.species, T, np.atleast_1d(P_pa), np.atleast_1d(PL_km), MF_VAL, MF_ID)
    if out is None:
        out = torch.empty((T.size, grid.n), dtype=torch.float32, device=device())
    voigt_sum(lines, grid, T, p_atm, w, out_f32=out)
    return out


def tud(OD, grid, T, Z, Altitudes=(500,), theta_r=0.0, N_angle=30, returnOD=False, per_angle=False, out=None):
    """tau[nAlt*nMu][n], Lu[nAlt*nMu][n], Ld[n] float32 device tensors from OD[nL][n] (rtx_tud).
    out=(tau, Lu, Ld): write into caller-owned tensors (tau/Lu [nAlt*nMu][>=n] with a common row stride),
    e.g. the rows of the packed block a wavenumber shard all-gathers."""
    lib = _lib.load()
    T = np.atleast_1d(np.asarray(T, dtype=np.float64))
    Z = np.atleast_1d(np.asarray(Z, dtype=np.float64))
    Z_s = np.array([Altitudes], dtype=np.float64).ravel()
    mu_s = np.array([1.0 / np.cos(np.asarray(theta_r, dtype=np.float64))], dtype=np.float64).ravel()
    nL = T.size
    assert OD.dtype == torch.float32 and OD.is_cuda and OD.dim() == 2 and OD.shape[0] == nL and OD.shape[1] >= grid.n
    assert OD.stride(1) == 1
    if mu_s.size > TUD_MAX_MU:
        # more slant paths than one launch takes (radiative_transfer.py:346-356 loops over any number): blocks of
        # TUD_MAX_MU, each a launch of its own; the downwelling (independent of mu) is simply recomputed
        if out is not None:
            raise ValueError("engine.tud: out= is limited to %d slant paths per call" % TUD_MAX_MU)
        th = np.asarray(theta_r, dtype=np.float64).ravel()
        tau = torch.empty((Z_s.size, mu_s.size, grid.n), dtype=torch.float32, device=OD.device)
        Lu = torch.empty_like(tau)
        Ld_ang = None
        for m0 in range(0, mu_s.size, TUD_MAX_MU):
            m1 = min(m0 + TUD_MAX_MU, mu_s.size)
            # the per-stream downwelling radiances do not depend on the slant paths: taken from the first block
            res = tud(OD, grid, T, Z, Altitudes=Altitudes, theta_r=th[m0:m1], N_angle=N_angle, returnOD=returnOD,
                      per_angle=per_angle and m0 == 0)
            t_c, l_c, Ld_c = res[:3]
            if m0 == 0:
                Ld = Ld_c
                Ld_ang = res[4] if per_angle else None
            tau[:, m0:m1] = t_c.view(Z_s.size, m1 - m0, grid.n)
            Lu[:, m0:m1] = l_c.view(Z_s.size, m1 - m0, grid.n)
        if per_angle:
            return tau.view(-1, grid.n), Lu.view(-1, grid.n), Ld, (Z_s.size, mu_s.size), Ld_ang
        return tau.view(-1, grid.n), Lu.view(-1, grid.n), Ld, (Z_s.size, mu_s.size)
    mask = np.ascontiguousarray(np.stack([(Z <= zs) for zs in Z_s]).astype(np.uint8))
    n_down = int(mask[-1].sum())  # quirk 3: nL is overwritten by the LAST altitude's count (:353, :370)
    dev = OD.device
    if out is None:
        tau = torch.empty((Z_s.size * mu_s.size, grid.n), dtype=torch.float32, device=dev)
        Lu = torch.empty_like(tau)
        Ld = torch.empty((grid.n,), dtype=torch.float32, device=dev)
    else:
        tau, Lu, Ld = out
        for t in (tau, Lu):
            assert t.dtype == torch.float32 and t.is_cuda and t.dim() == 2 and t.shape[0] == Z_s.size * mu_s.size
            assert t.shape[1] >= grid.n and t.stride(1) == 1 and t.stride(0) == tau.stride(0)
        assert Ld.dtype == torch.float32 and Ld.is_cuda and Ld.numel() >= grid.n and Ld.stride(0) == 1
    ld_out = tau.stride(0) if tau.shape[0] > 1 else max(tau.shape[1], grid.n)
    # rows of the per-stream output use the same leading dimension as tau / Lu (include/radtxfr_hip.h: [n_angle][ld_out])
    Ld_ang = torch.empty((int(N_angle), ld_out), dtype=torch.float32, device=dev) if per_angle else None
    T_h, T_p = _h(T)
    mu_h, mu_p = _h(mu_s)
    _lib.check(lib.rtx_tud(_ptr(OD), OD.stride(0), grid.byref(), nL, T_p, Z_s.size, mask.ctypes.data_as(C.c_void_p),
                           mu_s.size, mu_p, n_down, int(N_angle), int(bool(returnOD)), _ptr(tau), _ptr(Lu), _ptr(Ld),
                           _ptr(Ld_ang), ld_out, _stream_ptr()))
    if per_angle:
        return tau, Lu, Ld, (Z_s.size, mu_s.size), Ld_ang[:, :grid.n]
    return tau, Lu, Ld, (Z_s.size, mu_s.size)


class TudRunner:
    """compute_TUD for a stream of atmospheres on one spectral grid / line table / altitude grid (the reference's outer
    loop, Generate_LWIR_TUD.py:117-150): everything that does not depend on the atmosphere -- grid, prep object, sensor
    altitude masks, slant factors, device buffers, the ctypes argument objects -- is set up once; run() does the
    per-atmosphere host factors (TIPS ratios, column weights) and ONE call into the library (rtx_compute_tud: prologue +
    line-sum + TUD enqueued back to back). Outputs are float32 device tensors owned by the runner (or `out`), overwritten
    by the next run(): tau, Lu [nAlt*nMu][n], Ld [n], OD [nL][n]."""

    def __init__(self, lines, grid, Z, n_layers=None, Altitudes=(500,), theta_r=0.0, N_angle=30, returnOD=False, out=None,
                 OD=None, plan=None):
        self.lib = _lib.load()
        self.lines, self.grid = lines, grid
        Z = np.atleast_1d(np.asarray(Z, dtype=np.float64))
        self.nL = int(Z.size if n_layers is None else n_layers)
        Z_s = np.array([Altitudes], dtype=np.float64).ravel()
        self.mu = np.ascontiguousarray(np.array([1.0 / np.cos(np.asarray(theta_r, dtype=np.float64))], dtype=np.float64).ravel())
        if self.mu.size > TUD_MAX_MU:
            raise ValueError("TudRunner takes at most %d slant paths" % TUD_MAX_MU)
        self.mask = np.ascontiguousarray(np.stack([(Z <= zs) for zs in Z_s]).astype(np.uint8))
        self.n_down = int(self.mask[-1].sum())  # quirk 3: the LAST altitude's layer count (:353, :370)
        self.shape = (Z_s.size, self.mu.size)
        self.N_angle, self.returnOD = int(N_angle), int(bool(returnOD))
        dev = device()
        nrow = Z_s.size * self.mu.size
        self.OD = OD if OD is not None else torch.empty((self.nL, grid.n), dtype=torch.float32, device=dev)
        if out is None:
            self.tau = torch.empty((nrow, grid.n), dtype=torch.float32, device=dev)
            self.Lu = torch.empty_like(self.tau)
            self.Ld = torch.empty((grid.n,), dtype=torch.float32, device=dev)
        else:
            self.tau, self.Lu, self.Ld = out
        self.set_outputs(self.tau, self.Lu, self.Ld)
        # plan: a VoigtPlan of the caller's (two pipelines on one device must not share per-(line, layer) records);
        # default = the table's cached plan, shared by everything that runs on the device's current stream
        self.plan = plan if plan is not None else lines.plan(self.nL, grid.n)
        self._env = np.empty(2 * self.nL + 2 * len(lines.species) * self.nL + len(lines.species), dtype=np.float64)

    def set_outputs(self, tau, Lu, Ld):
        """Point the next run() at other output tensors (e.g. the other half of a double-buffered packed block)."""
        nrow = self.shape[0] * self.shape[1]
        for t in (tau, Lu):
            assert t.dtype == torch.float32 and t.is_cuda and t.dim() == 2 and t.shape[0] == nrow
            assert t.shape[1] >= self.grid.n and t.stride(1) == 1 and t.stride(0) == tau.stride(0)
        assert Ld.dtype == torch.float32 and Ld.is_cuda and Ld.numel() >= self.grid.n and Ld.stride(0) == 1
        self.tau, self.Lu, self.Ld = tau, Lu, Ld
        self._ld_out = tau.stride(0) if tau.shape[0] > 1 else max(tau.shape[1], self.grid.n)
        self._ptrs = (C.c_void_p(tau.data_ptr()), C.c_void_p(Lu.data_ptr()), C.c_void_p(Ld.data_ptr()))

    def run(self, T, P_pa, PL_km, MF_VAL, MF_ID, partitionFunction=None):
        nL, lines = self.nL, self.lines
        T = np.ascontiguousarray(T, dtype=np.float64)
        assert T.size == nL
        w, p_atm = layer_weights_od(lines.species, T, P_pa, PL_km, MF_VAL, MF_ID)
        qratio, mass = species_factors(lines.species, T, partitionFunction, weight=w)
        nS = len(lines.species)
        env = self._env  # T | p | qratio | weight | mass, one contiguous host block
        env[0:nL] = T
        env[nL:2 * nL] = p_atm
        env[2 * nL:2 * nL + nS * nL] = qratio.ravel()
        env[2 * nL + nS * nL:2 * nL + 2 * nS * nL] = w.ravel()
        env[2 * nL + 2 * nS * nL:] = mass
        base = env.ctypes.data
        vp = C.c_void_p
        with trace_range("rtx_compute_tud"):
            _lib.check(self.lib.rtx_compute_tud(
                self.plan._h, lines._h, self.grid.byref(), nL, vp(base), vp(base + 8 * nL), vp(base + 16 * nL),
                vp(base + 8 * (2 * nL + nS * nL)), vp(base + 8 * (2 * nL + 2 * nS * nL)), 1.0, 0.0, 0.0, 50.0, 0.0,
                self.shape[0], self.mask.ctypes.data_as(vp), self.shape[1], self.mu.ctypes.data_as(vp), self.n_down, self.N_angle,
                self.returnOD, vp(self.OD.data_ptr()), self.OD.stride(0), self._ptrs[0], self._ptrs[1], self._ptrs[2], self._ld_out,
                _stream_ptr()))
        return self.tau, self.Lu, self.Ld


class TudPipelines:
    """P independent TudRunners for a stream of atmospheres on one grid / table: each has its own HIP stream, per-(line,
    layer) records (VoigtPlan), optical-depth buffer and outputs, and atmosphere k runs on pipeline k mod P. Within a
    pipeline the four kernels of a step run back to back; across pipelines the fp64 prologue and the HBM-bound TUD pass of
    one atmosphere share the chip with the VALU-bound line-sum of the next (C3 on MI355X, bench.py: 1.97 / 1.90 / 1.90 /
    1.91 ms per atmosphere with 1 / 2 / 3 / 4 pipelines; profiles/r3_time_pipeline.txt). Results are the single-runner results bit for bit.
    outs: optional list of P (tau, Lu, Ld) output triples (e.g. rows of packed blocks that are all-gathered)."""

    def __init__(self, lines, grid, Z, n_layers=None, n_pipes=2, outs=None, **kw):
        self.streams = [torch.cuda.Stream() for _ in range(int(n_pipes))]
        self.runs = []
        nL = int(np.atleast_1d(Z).size if n_layers is None else n_layers)
        for p, s in enumerate(self.streams):
            with torch.cuda.stream(s):
                self.runs.append(TudRunner(lines, grid, Z, n_layers=nL, plan=VoigtPlan(lines, nL, grid.n),
                                           out=None if outs is None else outs[p], **kw))
        self.k = 0

    def run(self, T, P_pa, PL_km, MF_VAL, MF_ID):
        """Enqueue one atmosphere on the next pipeline; returns (pipeline index, (tau, Lu, Ld) of that pipeline)."""
        p = self.k % len(self.runs)
        self.k += 1
        with torch.cuda.stream(self.streams[p]):
            out = self.runs[p].run(T, P_pa, PL_km, MF_VAL, MF_ID)
        return p, out

    def close(self):
        torch.cuda.synchronize()
        for r in self.runs:
            r.plan.close()
        self.runs = []


def planck(X, T, wavelength=False, grid=None):
    """out[nx][nT] float64 device tensor (rtx_planck). X: device fp64 tensor, or None with a Grid."""
    lib = _lib.load()
    nx = grid.n if X is None else X.numel()
    out = torch.empty((nx, T.numel()), dtype=torch.float64, device=T.device)
    _lib.check(lib.rtx_planck(grid.byref() if grid is not None else None, _ptr(X), nx, _ptr(T), T.numel(),
                              int(bool(wavelength)), _ptr(out), _stream_ptr()))
    return out


def apparent_radiance(X, emis, Ts, tau, La, Ld, dT=None, return_Ls=False):
    """Device tensors in, L[nX][nE][nA][nT or 1] float32 out (rtx_apparent_radiance)."""
    lib = _lib.load()
    nX, nE = emis.shape
    nA = Ts.numel()
    nT = dT.numel() if dT is not None else 1
    L = torch.empty((nX, nE, nA, nT), dtype=torch.float32, device=emis.device)
    Ls = torch.empty_like(L) if return_Ls else None
    _lib.check(lib.rtx_apparent_radiance(_ptr(X), nX, _ptr(emis), nE, _ptr(Ts), nA, _ptr(tau), _ptr(La), _ptr(Ld),
                                         _ptr(dT), nT if dT is not None else 0, _ptr(L), _ptr(Ls), _stream_ptr()))
    return L, Ls


def ils(kind, Y, centre, sigma, X=None, grid=None):
    """Y[nx][nS] float32 device -> Y_out[nB][nS] float32 (rtx_ils). kind 0 triangle, 1 Gaussian."""
    lib = _lib.load()
    assert Y.dtype == torch.float32 and Y.is_cuda and Y.dim() == 2 and Y.stride(1) == 1
    nx, nS = Y.shape
    nB = centre.numel()
    out = torch.empty((nB, nS), dtype=torch.float32, device=Y.device)
    _lib.check(lib.rtx_ils(int(kind), grid.byref() if grid is not None else None, _ptr(X), nx, _ptr(Y), nS, Y.stride(0),
                           nB, _ptr(centre), _ptr(sigma), _ptr(out), _stream_ptr()))
    return out


def max_wing_cm(columns, T_layers, p_atm_layers, omega_wing=0.0, omega_wing_hw=50.0):
    """Upper bound [cm^-1] of OmegaWingF (misc/hapi.py:11131) over all lines and layers, plus the largest
    pressure shift: how far outside a wavenumber shard a line centre can sit and still contribute.
    Used to give each GPU only the lines that can reach its shard."""
    T = np.asarray(T_layers, dtype=np.float64)
    p = np.asarray(p_atm_layers, dtype=np.float64)
    nu = np.asarray(columns["nu"], dtype=np.float64)
    if nu.size == 0:
        return float(omega_wing)
    g = float(np.max(columns["gamma_air"]))
    n_hi, n_lo = float(np.max(columns["n_air"])), float(np.min(columns["n_air"]))
    tr = TREF / T
    g0 = g * np.max(p * np.maximum(tr ** n_hi, tr ** n_lo))
    gd = 3.6e-7 * float(np.max(nu)) * np.sqrt(float(np.max(T)) / 1.0)  # mass >= 1 g/mol: generous
    shift = float(np.max(np.abs(columns["delta_air"]))) * float(np.max(p))
    return max(float(omega_wing), omega_wing_hw * g0, omega_wing_hw * min(gd, 10.0 * g0 + 1.0)) + shift


# Relative cost of one line-sum tile, fitted (non-negative least squares) to the step time of 32 contiguous chunks of the
# C3 grid on MI355X (tools/shard_balance.py, profiles/r3_shard_balance.txt): per line whose window reaches the tile
# (classification + row-level evaluations) and per Weideman band row; the line centres' near-zone rows come out
# collinear with the reach count (0), and the per-(tile, layer) constant (zeroing, interpolation, stores, the TUD pass)
# cannot be told from the per-launch constant on equal chunks -- a small value keeps empty spans from costing nothing.
TILE_COST = {"tile": 100.0, "reach": 30.0, "centre": 0.0, "band_row": 24.0}


def tile_costs(columns, xmin, step, n_total, T_layers, p_atm_layers, tile, omega_wing=0.0, omega_wing_hw=50.0, coef=None):
    """Estimated line-sum + TUD cost of every `tile`-point tile of the axis xmin + i*step, i < n_total, summed over the
    layers: cost[n_tiles]. Host arithmetic on the table's columns only (windows W = max(OmegaWing, HW*gamma0, HW*gammaD),
    misc/hapi.py:11131; band half-width (15 - y)/cte where y < 15, :9840) -- the same on every rank. Used to cut
    wavenumber shards of equal COST rather than equal length (dist.tile_aligned_bounds)."""
    c = dict(TILE_COST)
    if coef:
        c.update(coef)
    T = np.atleast_1d(np.asarray(T_layers, dtype=np.float64))
    p = np.atleast_1d(np.asarray(p_atm_layers, dtype=np.float64))
    n_tiles = (int(n_total) + int(tile) - 1) // int(tile)
    cost = np.full(n_tiles, c["tile"] * T.size)
    nu = np.asarray(columns["nu"], dtype=np.float64)
    if nu.size == 0:
        return cost
    ga = np.asarray(columns["gamma_air"], dtype=np.float64)
    na = np.asarray(columns["n_air"], dtype=np.float64)
    M = np.asarray(columns["molec_id"]).astype(np.int64)
    I = np.asarray(columns["local_iso_id"]).astype(np.int64)
    mass = np.ones(nu.size)
    code = M * 1000003 + I
    for u in np.unique(code):
        try:
            mass[code == u] = tips.molecularMass(int(u // 1000003), int(u % 1000003))
        except Exception:
            pass  # unknown isotopologue: the default only skews the estimate
    span = float(step) * int(tile)
    tc = np.floor((nu - xmin) / span).astype(np.int64)  # tile of the line centre
    inside = (tc >= 0) & (tc < n_tiles)
    sqln2 = np.sqrt(np.log(2.0))
    for k in range(T.size):
        g0 = ga * p[k] * (TREF / T[k]) ** na
        gd = np.sqrt(2.0 * CBOLTS * T[k] * np.log(2.0) / (mass * 1.66053873e-27 * 1000.0) / 2.99792458e10 ** 2) * nu
        W = np.maximum(omega_wing, np.maximum(omega_wing_hw * g0, omega_wing_hw * gd))
        t_lo = np.clip(np.floor((nu - W - xmin) / span), 0, n_tiles).astype(np.int64)
        t_hi = np.clip(np.floor((nu + W - xmin) / span) + 1, 0, n_tiles).astype(np.int64)
        d = np.bincount(t_lo, minlength=n_tiles + 1)[:n_tiles + 1] - np.bincount(t_hi, minlength=n_tiles + 1)[:n_tiles + 1]
        cost += c["reach"] * np.cumsum(d)[:n_tiles]
        y = g0 * sqln2 / gd
        rows = np.where(y < 15.0, 2.0 * (15.0 - y) * gd / sqln2 / (64.0 * step) + 1.0, 0.0)
        cost += np.bincount(tc[inside], weights=(c["centre"] + c["band_row"] * rows)[inside], minlength=n_tiles)[:n_tiles]
    return cost


# ---- post-processing: smooth / reduceResolution (rtx_fir_reflect, rtx_cubic_resample) ------------
_WINDOWS = ("flat", "hanning", "hamming", "bartlett", "blackman")


def window_taps(window_len, window="hanning", symmetric=False):
    """FIR taps and centre of radiative_transfer.smooth (:1314-1324) in the convention of rtx_fir_reflect:
    out[i] = sum_k taps[k] * x[R(i + k - centre)]. symmetric=True gives reduceResolution's symmetrised smoother
    (:1331), the mean of smoothing x and smoothing x reversed."""
    wl = int(window_len)
    w = np.ones(wl, "d") if window == "flat" else getattr(np, window)(wl)
    w = w / w.sum()
    ix0 = int(np.ceil(wl / 2 - 1))
    c = wl - 1 - ix0  # out[i] = sum_q w[wl-1-q] * s[i + ix0 + q],  s[k] = x[R(k - (wl - 1))]
    fwd = w[::-1].copy()
    if not symmetric:
        return fwd, c
    # the reversed pass uses offsets -(k - c): offsets d in [-(wl-1-c), c]
    m = max(c, wl - 1 - c)
    taps = np.zeros(2 * m + 1)
    for k in range(wl):
        taps[m + (k - c)] += 0.5 * fwd[k]
        taps[m - (k - c)] += 0.5 * fwd[k]
    return taps, m


def fir_reflect(Y, taps, centre):
    """Y [rows][n] float32/float64 device tensor -> [rows][n] float64 (rtx_fir_reflect)."""
    lib = _lib.load()
    assert Y.is_cuda and Y.dim() == 2 and Y.stride(1) == 1 and Y.dtype in (torch.float32, torch.float64)
    rows, n = Y.shape
    taps = np.ascontiguousarray(taps, dtype=np.float64)
    out = torch.empty((rows, n), dtype=torch.float64, device=Y.device)
    ld = Y.stride(0) if rows > 1 else n  # the stride of a size-1 dimension is arbitrary
    _lib.check(lib.rtx_fir_reflect(_ptr(Y), int(Y.dtype == torch.float64), ld, rows, n, taps.ctypes.data, taps.size,
                                   int(centre), _ptr(out), out.stride(0), _stream_ptr()))
    return out


def cubic_resample(Ysm, x0, h, x_out, checked=True):
    """Cubic spline through every sample of the uniform axis x0 + i*h, at x_out (device fp64): [rows][n_out] fp64.
    checked=False: the caller has verified the range of x_out on the host; no device read-back, no synchronisation."""
    lib = _lib.load()
    assert Ysm.is_cuda and Ysm.dtype == torch.float64 and Ysm.dim() == 2 and Ysm.stride(1) == 1
    assert x_out.is_cuda and x_out.dtype == torch.float64 and x_out.dim() == 1 and x_out.is_contiguous()
    rows, n = Ysm.shape
    out = torch.empty((rows, x_out.numel()), dtype=torch.float64, device=Ysm.device)
    fn = lib.rtx_cubic_resample if checked else lib.rtx_cubic_resample_unchecked
    _lib.check(fn(_ptr(Ysm), Ysm.stride(0) if rows > 1 else n, rows, n, float(x0), float(h), _ptr(x_out), x_out.numel(),
                  _ptr(out), out.stride(0), _stream_ptr()))
    return out


SPL_END_GUARD = 24   # knots between the last output of an end region and the cut of its local spline (0.268^24 = 2e-14)
SPL_END_MAX = 768    # local knots rtx_cubic_end takes


def _smoothed_axis_ends(x0, h, n, taps, c, m):
    """The first and last m values of the reference's smoothed axis X_ = sm(X) (radiative_transfer.py:1331-1334): the
    symmetrised window applied to X = x0 + i*h with smooth()'s reflection padding. Host arithmetic on the axis only."""
    k = np.arange(taps.size)
    i = np.arange(m)
    j = i[:, None] + k[None, :] - c
    j = np.where(j < 0, -j, j)
    lo = (taps[None, :] * (x0 + h * j)).sum(axis=1)
    j = (n - m + i)[:, None] + k[None, :] - c
    j = np.where(j >= n, 2 * (n - 1) - j, j)
    hi = (taps[None, :] * (x0 + h * j)).sum(axis=1)
    return lo, hi


def _reduce_plan(x0, h, n, dX, N, window, x_out, device):
    """Everything of reduceResolution that does not depend on the spectra: output axis, window taps, the split of the
    outputs into a low-end region, the interior and a high-end region, and the true knots of the two end regions."""
    sm_factor = int(np.round(dX / h))
    if sm_factor < 3:
        raise ValueError(f"reduceResolution: dX/dX_in rounds to {sm_factor}; the window needs at least 3 samples")
    if window not in _WINDOWS:
        raise ValueError(f"window must be one of {_WINDOWS}")
    taps, c = window_taps(sm_factor, window, symmetric=True)
    if x_out is None:
        # the reference's default axis runs from X_[smFactor] to X_[-smFactor-1] (:1336-1338); a window length inside, the
        # smoothed axis is the axis itself
        xa, xb = x0 + sm_factor * h, x0 + (n - sm_factor - 1) * h
        v = N * (xb - xa) / dX
        x_out = np.linspace(xa, xb, int(np.ceil(v - 1e-9 * max(1.0, abs(v)))) + 1)  # see the shim's docstring: rounding-proof ceil
    x_out = np.ascontiguousarray(x_out, dtype=np.float64)
    # The interior evaluation (uniform knots, cardinal spline) is offered where the reflection padding of smooth() and the
    # not-a-knot end condition have faded below 1e-11: `margin` samples inside. Outputs nearer an end go through the local
    # not-a-knot spline on the true knots (rtx_cubic_end).
    margin = max((sm_factor + 1) // 2 + 20, 27)  # (27: what rtx_cubic_resample itself asks of its abscissae)
    x_lo, x_hi = x0 + margin * h, x0 + (n - 1 - margin) * h
    n_lo = n_hi = 0
    ends = None
    if x_out.size and (x_out.min() < x_lo or x_out.max() > x_hi):
        if np.any(np.diff(x_out) < 0):
            raise NotImplementedError("reduceResolution: X_out must be ascending when it reaches into the end regions of the axis")
        n_lo = int(np.searchsorted(x_out, x_lo, side="left"))
        n_hi = int(x_out.size - np.searchsorted(x_out, x_hi, side="right"))
        if n_lo + n_hi > x_out.size:  # a short axis: everything is an end region; the low end takes the lower half
            n_lo = int(np.searchsorted(x_out, x0 + 0.5 * (n - 1) * h, side="left"))
            n_hi = x_out.size - n_lo
        m = min(n, margin + SPL_END_GUARD + 2 + (sm_factor + 1) // 2)
        if m > SPL_END_MAX:
            raise NotImplementedError(
                f"reduceResolution: output points within {margin} samples of an end of the input axis need a local spline of {m} "
                f"knots (window {sm_factor}); supported up to {SPL_END_MAX}")
        k_lo, k_hi = _smoothed_axis_ends(x0, h, n, taps, c, m)
        if (n_lo and x_out[0] < k_lo[0]) or (n_hi and x_out[-1] > k_hi[-1]):
            raise NotImplementedError("reduceResolution: X_out outside the smoothed axis (extrapolation is not supported)")
        if m < n and ((n_lo and x_out[n_lo - 1] > k_lo[m - 1 - SPL_END_GUARD]) or (n_hi and x_out[x_out.size - n_hi] < k_hi[SPL_END_GUARD])):
            raise NotImplementedError("reduceResolution: the axis is too short for its end regions to be treated separately")
        ends = (m, torch.as_tensor(k_lo, device=device), torch.as_tensor(k_hi, device=device))
    return x_out, torch.as_tensor(x_out, device=device), taps, c, n_lo, n_hi, ends


def _reduce_apply(Y, x0, h, n, plan, checked):
    x_out, x_dev, taps, c, n_lo, n_hi, ends = plan
    Ysm = fir_reflect(Y, taps, c)
    n_out = x_out.size
    n_mid = n_out - n_lo - n_hi
    if not (n_lo or n_hi):
        return x_out, cubic_resample(Ysm, x0, h, x_dev, checked=checked)
    lib = _lib.load()
    rows = Ysm.shape[0]
    out = torch.empty((rows, n_out), dtype=torch.float64, device=Y.device)
    ld = Ysm.stride(0) if rows > 1 else n
    m, k_lo, k_hi = ends
    if n_lo:
        _lib.check(lib.rtx_cubic_end(_ptr(Ysm), ld, rows, 0, m, 0, _ptr(k_lo), C.c_void_p(x_dev.data_ptr()), n_lo,
                                     C.c_void_p(out.data_ptr()), out.stride(0), _stream_ptr()))
    if n_mid:
        out[:, n_lo:n_lo + n_mid] = cubic_resample(Ysm, x0, h, x_dev[n_lo:n_lo + n_mid], checked=checked)
    if n_hi:
        _lib.check(lib.rtx_cubic_end(_ptr(Ysm), ld, rows, n - m, m, 1, _ptr(k_hi), C.c_void_p(x_dev.data_ptr() + 8 * (n_out - n_hi)), n_hi,
                                     C.c_void_p(out.data_ptr() + 8 * (n_out - n_hi)), out.stride(0), _stream_ptr()))
    return x_out, out


def reduce_resolution(Y, x0, h, n, dX, N=4, window="hanning", x_out=None):
    """Device-resident reduceResolution: Y [rows][n] (float32 as rtx_tud writes it, or float64) on the uniform axis
    x0 + i*h -> (x_out host fp64, Y_out [rows][n_out] fp64 device). See radiative_transfer.reduceResolution."""
    return _reduce_apply(Y, x0, h, n, _reduce_plan(x0, h, n, dX, N, window, x_out, Y.device), checked=True)


_REDUCE_PLANS = {}


def reduce_resolution_cached(Y, x0, h, n, dX, N=4, window="hanning"):
    """reduce_resolution on its default output axis, for a stream of spectra on one grid (compute_TUD_batch: every
    atmosphere of a batch is reduced the same way): the output axis (host and device copies), the symmetrised window taps
    (a Python loop over the window), the end-region knots and the range checks are made once per (axis, dX, N, window,
    device) instead of per spectrum -- together they cost more host time than the device needs for the whole atmosphere --
    and the resampling runs without its device read-back."""
    key = (float(x0), float(h), int(n), float(dX), int(N), window, Y.device.index)
    plan = _REDUCE_PLANS.get(key)
    if plan is None:
        if len(_REDUCE_PLANS) > 8:
            _REDUCE_PLANS.clear()
        plan = _REDUCE_PLANS[key] = _reduce_plan(x0, h, n, dX, N, window, None, Y.device)
    return _reduce_apply(Y, x0, h, n, plan, checked=False)
