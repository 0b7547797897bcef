"""Drop-in for the hot-path functions of the reference's radiative_transfer.py.

Same names, positional order, keyword names and return shapes (spectral axis first) as the
reference (radiative_transfer.py:22-25, signatures listed in SURVEY.md section 8b):

    planckian                         :792-848
    make_spectral_axis                :251-271
    compute_OD                        :395-456    (LBLRTM replaced, see below)
    compute_TUD                       :274-392
    compute_LWIR_apparent_radiance    :1017-1069
    ILS_MAKO                          :1072-1263
    options, StdAtmos                 :75-183

NumPy in -> NumPy out (float64, like the reference); torch CUDA tensors in -> torch out where the
function is elementwise. All arithmetic runs in libradtxfr_hip.so on the GPU; there is no CPU path.

Conscious divergences from the reference:
  * compute_OD: the reference shells out to the LBLRTM Fortran binary with the AER line file; both
    are git-LFS stubs. Here OD(nu) = sum_m k_m(nu; T, p) x_m PL from the Voigt line-sum over the line
    table named by opts["line_table"] (a table in radtxfr_amd.hapi.LOCAL_TABLE_CACHE, or a column
    dict). No continuum. Chunking / TAPE5 / TAPE12 plumbing does not exist.
  * options are copied per call: kwargs no longer leak into the module-level default (quirk 2).
  * make_spectral_axis casts the point count to int (quirk 1: the reference crashes on NumPy>=1.18).
  * spectra are computed in float32 on the device and returned as float64.
"""
import os
import threading
import time

import numpy as np
import torch

from . import _hostio, engine
from . import hapi as _hapi

# Module constants (radiative_transfer.py:71-72)
c1 = 1.19104295315e-16  # [J m^2 / s]
c2 = 1.43877736830e-02  # [m K]

# 66-layer 1976 US standard atmosphere. The reference embeds a rounded copy (:77-144) of its
# StandardAtmosphere.csv (2 decimals for Z/PL/P/T, %.3E for the mixing ratios); this package ships
# the CSV (input fixture) and applies the same rounding, which reproduces the embedded table exactly
# (tests/test_host.py checks it against a fixture captured from the reference).
_CSV = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "StandardAtmosphere.csv")
StdAtmosCSV = np.loadtxt(_CSV, delimiter=",", skiprows=1)
StdAtmos = StdAtmosCSV.copy()
StdAtmos[:, 1:6] = np.round(StdAtmosCSV[:, 1:6], 2)
StdAtmos[:, 6:] = np.vectorize(lambda v: float("%.3E" % v))(StdAtmosCSV[:, 6:])

options = {
    "DVOUT": 0.0005,  # [cm^{-1}]
    "T": 296.0, "P": 101325.0, "PL": 1.0,
    "MF_ID": np.array([]), "MF_VAL": np.array([]),
    "line_table": None,  # replaces "LBLRTM"/"TAPE3": name in hapi.LOCAL_TABLE_CACHE or column dict
    "copy_axis": False,  # True: compute_TUD returns a fresh writable X (the reference's behaviour) instead of a cached read-only one
    "chunks": None,      # compute_TUD: wavenumber chunks whose device-to-host copies overlap the next chunk's kernels (None: automatic)
    # options for compute_TUD (:172-182)
    "Zs": StdAtmos[:, 1], "Ts": StdAtmos[:, 5], "Ps": StdAtmos[:, 4], "PLs": StdAtmos[:, 3],
    "MFs_VAL": StdAtmos[:, 6:14] * 1e6, "MFs_ID": np.array([1, 2, 3, 4, 5, 6, 7, 22]),
    "theta_r": 0, "N_angle": 30, "Altitudes": np.asarray([500]), "save": False, "returnOD": False,
}


def _is_torch(x):
    return isinstance(x, torch.Tensor)


def _dev_f64(x, dev):
    if _is_torch(x):
        return x.to(device=dev, dtype=torch.float64).contiguous()
    a = np.asarray(x, dtype=np.float64)
    return torch.as_tensor(a if a.flags.writeable else a.copy(), device=dev).contiguous()  # read-only views (broadcast_to)


def _dev_f32(x, dev):
    if _is_torch(x):
        return x.to(device=dev, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float32), device=dev).contiguous()


def make_spectral_axis(Xmin, Xmax, DVOUT):
    """:251-271. Spacing is (Xmax-Xmin)/(nX-1), not DVOUT, exactly like the reference."""
    nX = int(np.ceil((Xmax - Xmin) / DVOUT))
    return np.linspace(Xmin, Xmax, nX)


def planckian(X_in, T_in, wavelength=False):
    """Planck spectral radiance, :792-848. Output shape (X.size, *T.shape);
    [uW/(cm^2 sr cm^-1)] or, for wavelength input in um, [uW/(cm^2 sr um)]."""
    as_torch = _is_torch(X_in) or _is_torch(T_in)
    dev = engine.device()
    X = _dev_f64(X_in, dev).flatten()
    T = _dev_f64(T_in, dev)
    dimsT = tuple(T.shape)
    if wavelength or float(X.mean()) < 50:
        if not wavelength:
            print("Assumes X given in µm; returning L in µF")
        wavelength = True
    L = engine.planck(X, T.flatten(), wavelength=wavelength).reshape((X.numel(), *dimsT))
    return L if as_torch else L.cpu().numpy()


def _planck_family(fn_name, X_in, V_in, wavelength, bad_value, spectral_dim, notice):
    """Shared body of brightnessTemperature / BT2L: the reference's reshaping rules (:890-907, :925-931)
    around one elementwise fp64 kernel."""
    import ctypes as C
    from . import _lib
    as_torch = _is_torch(X_in) or _is_torch(V_in)
    dev = engine.device()
    X = _dev_f64(X_in, dev).flatten()
    V = _dev_f64(V_in, dev)
    if spectral_dim != 0:
        V = V.swapaxes(0, spectral_dim)
    if V.dim() == 1:
        V = V[:, None]
    dims = tuple(V.shape)
    V2 = V.reshape(dims[0], -1).contiguous()
    if V2.shape[0] != X.numel():
        raise ValueError("operands could not be broadcast together with shapes (%d,1) (%d,%d)" % (X.numel(), *V2.shape))
    if wavelength or float(X.mean()) < 50:
        if not wavelength:
            print(notice)
        wavelength = True
    out = torch.empty_like(V2)
    lib = _lib.load()
    _lib.check(getattr(lib, fn_name)(C.c_void_p(X.data_ptr()), X.numel(), C.c_void_p(V2.data_ptr()), V2.shape[1],
                                     int(bool(wavelength)), float(bad_value), C.c_void_p(out.data_ptr()),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    out = out.reshape((X.numel(), *dims[1:]))  # a 1-D input comes back (nX, 1), as in the reference
    if spectral_dim != 0:
        out = out.swapaxes(0, spectral_dim)
    return out if as_torch else out.cpu().numpy()


def brightnessTemperature(X_in, L_in, wavelength=False, bad_value=np.nan, spectral_dim=0):
    """Brightness temperature [K] of spectral radiance L, signature of :851-933. X (nX,), L with the
    spectral axis at `spectral_dim`; unphysical radiances (non-finite or <= 0) give `bad_value` (:922-923)."""
    return _planck_family("rtx_brightness_temperature", X_in, L_in, wavelength, bad_value, spectral_dim,
                          "Assumes X given in µm and L given in µF")


def BT2L(X_in, T_in, wavelength=False, bad_value=np.nan, spectral_dim=0):
    """Spectral radiance of a brightness-temperature spectrum, signature of :936-1014."""
    return _planck_family("rtx_bt2l", X_in, T_in, wavelength, bad_value, spectral_dim,
                          "Assumes X given in µm and L given in µF")


def _resolve_table(spec):
    if spec is None:
        raise Exception("compute_OD: set opts['line_table'] to a table in radtxfr_amd.hapi.LOCAL_TABLE_CACHE "
                        "(the reference's LBLRTM binary and AER TAPE3 are git-LFS stubs; there is no built-in line file)")
    if isinstance(spec, engine.LineTable):
        return spec
    if isinstance(spec, str):
        if spec not in _hapi.LOCAL_TABLE_CACHE:
            raise Exception("%s: no such table. Check tableList() for more info." % spec)
        return _hapi._device_table([spec])
    if isinstance(spec, dict):
        key = "__dict_%d" % id(spec)
        if key not in _hapi.LOCAL_TABLE_CACHE or _hapi.LOCAL_TABLE_CACHE[key].get("_src") is not spec:
            _hapi.LOCAL_TABLE_CACHE[key] = {"header": {"number_of_rows": len(spec["nu"])}, "data": spec, "_src": spec}
        return _hapi._device_table([key])
    raise TypeError("opts['line_table'] must be a table name, a column dict or an engine.LineTable")


def compute_OD(Xmin_in, Xmax_in, opts=options, **kwargs):
    """Monochromatic optical depth of one homogeneous layer, signature of :395-456.
    kwargs honoured: T [K], P [Pa], PL [km], MF_VAL [ppmv], MF_ID (HITRAN ids), DVOUT, line_table.
    Returns (X_out, OD_out)."""
    o = dict(opts)
    o.update(kwargs)
    DVOUT = o.get("DVOUT", 0.025)
    X = make_spectral_axis(Xmin_in, Xmax_in, DVOUT)
    tbl = _resolve_table(o.get("line_table"))
    grid = engine.Grid(Xmin_in, Xmax_in, X.size)
    OD = engine.optical_depths(tbl, grid, [o["T"]], [o["P"]], [o["PL"]], np.asarray(o["MF_VAL"], dtype=np.float64)[None, :],
                               o["MF_ID"])
    return X, OD[0].double().cpu().numpy()


_AXIS_CACHE = {}


def _cached_axis(Xmin, Xmax, DVOUT):
    """make_spectral_axis with a two-entry cache: np.linspace of the 5.5 M-point C3 axis costs ~12 ms of host time per call,
    more than the whole device computation. The cached array is returned READ-ONLY (every call on the same grid hands out the
    same object); callers that want to modify it copy it."""
    key = (float(Xmin), float(Xmax), float(DVOUT))
    X = _AXIS_CACHE.get(key)
    if X is None:
        X = make_spectral_axis(Xmin, Xmax, DVOUT)
        X.setflags(write=False)
        if len(_AXIS_CACHE) >= 2:
            _AXIS_CACHE.pop(next(iter(_AXIS_CACHE)))
        _AXIS_CACHE[key] = X
    return X


_STAGING = {}
_TRACE = float(os.environ.get("RADTXFR_TRACE", "0") or 0)  # developer aid: print the phases of slow compute_TUD calls


def _rows_to_host_f64(rows, stream=None):
    """float32 device rows [(k_i, n)] -> float64 NumPy arrays (see _hostio): zero-copy views of a pinned block while the
    pinned bytes lent out stay under _hostio.PINNED_RESULT_CAP, else fresh pageable arrays filled from a reusable pinned
    staging ring by the host thread pool. Returns (arrays, event): the arrays are valid once `event` has completed
    (event.synchronize()); on the pageable path they are complete on return."""
    got = _hostio.rows_to_pinned_f64(rows, stream)
    if got is not None:
        return got
    key = (rows[0].device.index, threading.get_ident())
    st = _STAGING.get(key)
    if st is None:
        st = _STAGING[key] = _hostio.Staging(depth=1)
    ticket = st.stage(rows, stream)
    return st.collect(ticket), ticket[3]


def _tud_shapes(tau2, Lu2, nZ, nMu):
    """The reference's squeeze rules (:357-365) on [nZ*nMu][nX] host rows."""
    tau_ = tau2.reshape(nZ, nMu, -1).transpose(2, 0, 1)
    Lu_ = Lu2.reshape(nZ, nMu, -1).transpose(2, 0, 1)
    if nZ == 1 and nMu == 1:
        return tau_[:, 0, 0], Lu_[:, 0, 0]
    if nZ == 1:
        return np.ascontiguousarray(tau_[:, 0, :]), np.ascontiguousarray(Lu_[:, 0, :])
    if nMu == 1:
        return np.ascontiguousarray(tau_[:, :, 0]), np.ascontiguousarray(Lu_[:, :, 0])
    return np.ascontiguousarray(tau_), np.ascontiguousarray(Lu_)


_SIDE_STREAMS = {}


def _compute_tud_chunked(tbl, grid, Z, T, P, PL, MF, ID, Z_s, theta, nA, returnOD, n_chunks):
    """compute_TUD's device work in n_chunks tile-aligned wavenumber chunks, each chunk's widening and device-to-host copy
    (side stream) under the next chunk's kernels: a single call is PCIe-bound (132 MB of float64 at C3 size: 2.7 ms against
    1.9 ms of kernels), and chunks cut on line-sum tile boundaries give the unchunked bits. Returns (tau_h, Lu_h, Ld_h,
    (nZ, nMu)) as float64 NumPy views of one pinned block, or None when no pinned block can be lent (the caller then takes
    the plain path)."""
    from . import _lib, dist as _dist
    n = grid.n
    tile = int(_lib.load().rtx_voigt_tile_points())
    offs = _dist.tile_aligned_bounds(n, n_chunks, tile)
    nL = T.size
    nZ, nMu = Z_s.size, np.atleast_1d(theta).size
    nrow = nZ * nMu
    lent = _hostio.lend_block((2 * nrow + 1, n))
    if lent is None:
        return None
    host, arr = lent
    dev = engine.device()
    tau = torch.empty((nrow, n), dtype=torch.float32, device=dev)
    Lu = torch.empty_like(tau)
    Ld = torch.empty((n,), dtype=torch.float32, device=dev)
    max_len = int(np.diff(offs).max())
    OD = torch.empty((nL, max_len), dtype=torch.float32, device=dev)
    plan = tbl.plan(nL, n)
    side = _SIDE_STREAMS.get(dev.index)
    if side is None:
        side = _SIDE_STREAMS[dev.index] = torch.cuda.Stream()
    done = None
    for c in range(n_chunks):
        off, ln = int(offs[c]), int(offs[c + 1] - offs[c])
        if ln == 0:
            continue
        sl = slice(off, off + ln)
        run = engine.TudRunner(tbl, grid.shard(grid.offset + off, ln), Z, n_layers=nL, Altitudes=Z_s, theta_r=theta, N_angle=nA,
                               returnOD=returnOD, out=(tau[:, sl], Lu[:, sl], Ld[sl]), OD=OD[:, :ln], plan=plan)
        run.run(T, P, PL, MF, ID)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            blk = torch.empty((2 * nrow + 1, ln), dtype=torch.float64, device=dev)  # widened on the device
            blk[:nrow].copy_(tau[:, sl])
            blk[nrow:2 * nrow].copy_(Lu[:, sl])
            blk[2 * nrow].copy_(Ld[sl])
            for r in range(2 * nrow + 1):
                host[r, sl].copy_(blk[r], non_blocking=True)  # contiguous row segments
            done = torch.cuda.Event()
            done.record(side)
    for t in (tau, Lu, Ld, OD):
        t.record_stream(side)
    if done is not None:
        done.synchronize()
    return arr[:nrow], arr[nrow:2 * nrow], arr[2 * nrow:], (nZ, nMu)


def compute_TUD(Xmin, Xmax, opts=options, **kwargs):
    """Monochromatic transmittance, upwelling and downwelling radiance, signature of :274-392.

    kwargs honoured (:304-314): DVOUT, Zs, Ts, Ps, PLs, MFs_VAL, MFs_ID, theta_r, N_angle, Altitudes,
    save, returnOD, plus line_table. Returns (X, tau, Lu, Ld); tau/Lu are (nX,), (nX,nMu), (nX,nZ) or
    (nX,nZ,nMu) by the reference's squeeze rules (:357-365); Ld is (nX,).
    Quirks 3-6 of SURVEY.md section 9 are reproduced (downwelling uses the layer count of the last
    sensor altitude; tau uses the Z<=zs mask, L-up the first count layers; returnOD; theta=0 weight 0).
    X is a cached read-only array (see _cached_axis; copy_axis=True gives the reference's fresh writable one); the
    spectra are fresh float64 arrays -- views of page-locked memory while less than _hostio.PINNED_RESULT_CAP is lent
    out to live results, ordinary pageable arrays beyond that.
    """
    trace = [time.perf_counter()] if _TRACE else None
    o = dict(opts)
    o.update(kwargs)
    Z = np.asarray(o["Zs"], dtype=np.float64)
    T = np.asarray(o["Ts"], dtype=np.float64)
    P = np.asarray(o["Ps"], dtype=np.float64)
    PL = np.asarray(o["PLs"], dtype=np.float64)
    MF = np.asarray(o["MFs_VAL"], dtype=np.float64)
    ID = np.asarray(o["MFs_ID"])
    nA = int(o["N_angle"])
    f = lambda x: np.array([x]).ravel()
    Z_s = f(o["Altitudes"])
    mu_s = f(1.0 / np.cos(o["theta_r"]))
    X_ = _cached_axis(Xmin, Xmax, o["DVOUT"])
    grid = engine.Grid(Xmin, Xmax, X_.size)
    if trace:
        trace.append(time.perf_counter())
    tbl = _resolve_table(o.get("line_table"))
    if trace:
        trace.append(time.perf_counter())
    if o.get("copy_axis"):
        X_ = _hostio.copy_threaded(X_)
    chunks = o.get("chunks")
    if chunks is None:  # automatic: worth it once the copy of the result outweighs a chunk's fixed costs
        chunks = 4 if grid.n >= 2000000 else 2 if grid.n >= 500000 else 1
    if chunks > 1 and mu_s.size <= engine.TUD_MAX_MU and not o["save"]:
        got = _compute_tud_chunked(tbl, grid, Z, T, P, PL, MF, ID, Z_s, np.asarray(o["theta_r"], dtype=np.float64), nA,
                                   bool(o["returnOD"]), int(chunks))
        if got is not None:
            tau_h, Lu_h, Ld_h, (nZ, nMu) = got
            if trace:
                trace.append(time.perf_counter())
                if trace[-1] - trace[0] > _TRACE * 1e-3:
                    print("compute_TUD %.1f ms (chunked x%d): options+axis %.2f  table %.2f  kernels + copies %.2f"
                          % tuple([1e3 * (trace[-1] - trace[0]), chunks] + [1e3 * (b - a) for a, b in zip(trace, trace[1:])]), flush=True)
            tau_, Lu_ = _tud_shapes(tau_h, Lu_h, nZ, nMu)
            return X_, tau_, Lu_, Ld_h[0]
    if mu_s.size <= engine.TUD_MAX_MU and not o["save"]:
        # the common case: one call into the library (rtx_compute_tud)
        run = engine.TudRunner(tbl, grid, Z, n_layers=T.size, Altitudes=Z_s, theta_r=np.asarray(o["theta_r"], dtype=np.float64),
                               N_angle=nA, returnOD=bool(o["returnOD"]))
        tau, Lu, Ld = run.run(T, P, PL, MF, ID)
        nZ, nMu = run.shape
        OD = run.OD
    else:
        OD = engine.optical_depths(tbl, grid, T, P, PL, MF, ID)  # [nL][nX] float32 on the device
        res = engine.tud(OD, grid, T, Z, Altitudes=Z_s, theta_r=np.asarray(o["theta_r"], dtype=np.float64), N_angle=nA,
                         returnOD=bool(o["returnOD"]), per_angle=bool(o["save"]))
        tau, Lu, Ld, (nZ, nMu) = res[:4]
    if trace:
        trace.append(time.perf_counter())
    (tau_h, Lu_h, Ld_h), done = _rows_to_host_f64([tau, Lu, Ld[None, :]])
    if trace:
        trace.append(time.perf_counter())
    done.synchronize()
    if trace:
        trace.append(time.perf_counter())
        if trace[-1] - trace[0] > _TRACE * 1e-3:  # RADTXFR_TRACE=<ms>: phases of every call slower than that
            print("compute_TUD %.1f ms: options+axis %.2f  table %.2f  enqueue kernels %.2f  enqueue copies %.2f  wait %.2f"
                  % tuple(1e3 * v for v in [trace[-1] - trace[0]] + [b - a for a, b in zip(trace, trace[1:])]), flush=True)
    tau_, Lu_ = _tud_shapes(tau_h, Lu_h, nZ, nMu)
    Ld_ = Ld_h[0]
    if o["save"]:
        angles = np.linspace(0, np.pi / 2.0, nA, endpoint=False)
        t3 = tau_h.reshape(nZ, nMu, -1).transpose(2, 0, 1)
        u3 = Lu_h.reshape(nZ, nMu, -1).transpose(2, 0, 1)
        np.savez("ComputeTUD.npz", OD=OD.double().cpu().numpy().T, B=planckian(X_, T), tau=t3, Ld=res[4].double().cpu().numpy().T,
                 Lu=u3, X=X_, angles=angles, Z_s=Z_s, mu_s=mu_s)
    return X_, tau_, Lu_, Ld_


class _TudPipeline:
    """One device's share of compute_TUD_batch: the line table's copy on that device, two runners (two sets of device
    outputs: the copy of one is in flight while the other is being computed), each on a compute stream of its own, a
    side stream for the device-to-host copies, a pinned staging ring."""

    def __init__(self, dev, tbl, grid, Z, nL, Z_s, theta_r, N_angle, returnOD):
        self.dev = int(dev)
        with torch.cuda.device(self.dev):
            self.lines = tbl.on_device(self.dev)
            self.side = torch.cuda.Stream()
            # two runners, each with a stream, per-(line, layer) records and an optical-depth buffer of its own (another
            # pipeline may share the device and the table's cached plan): the prologue and TUD pass of one atmosphere overlap
            # the line-sum of the next (engine.TudPipelines)
            self.computes = [torch.cuda.Stream(), torch.cuda.Stream()]
            self.plans = [engine.VoigtPlan(self.lines, nL, grid.n) for _ in range(2)]
            self.runs = []
            for st, pl in zip(self.computes, self.plans):
                with torch.cuda.stream(st):
                    self.runs.append(engine.TudRunner(self.lines, grid, Z, n_layers=nL, Altitudes=Z_s, theta_r=theta_r, N_angle=N_angle,
                                                      returnOD=returnOD, plan=pl))
        self.busy = [None, None]  # copy-done event of each runner's outputs
        self.staging = _hostio.Staging(depth=2)
        self.k = 0

    def enqueue(self, a, ID, grid, x0, reduce, full_dtype):
        """Kernels + device-to-host copy of one atmosphere, all asynchronous. Returns a ticket for finish()."""
        j = self.k & 1
        self.k += 1
        run = self.runs[j]
        with torch.cuda.device(self.dev), torch.cuda.stream(self.computes[j]):
            if self.busy[j] is not None:
                self.computes[j].wait_event(self.busy[j])  # its previous outputs have left the device
            tau, Lu, Ld = run.run(np.asarray(a["Ts"], dtype=np.float64), np.asarray(a["Ps"], dtype=np.float64),
                                  np.asarray(a["PLs"], dtype=np.float64), np.asarray(a["MFs_VAL"], dtype=np.float64), ID)
            Xr = None
            if reduce is not None:
                rows = torch.cat([tau, Lu, Ld[None, :]])
                Xr, red = engine.reduce_resolution_cached(rows, x0, grid.step, grid.n, float(reduce["dX"]), N=reduce.get("N", 4),
                                                          window=reduce.get("window", "hanning"))
                nr = tau.shape[0]
                t = self.staging.stage([red[:nr], red[nr:2 * nr], red[2 * nr:]], stream=self.side)
            else:
                # full float64 spectra: zero-copy pinned blocks while the module's cap allows (a short batch then costs no host
                # work at all), the pinned staging ring + threaded widening into pageable arrays beyond it
                got = _hostio.rows_to_pinned_f64([tau, Lu, Ld[None, :]], stream=self.side) if full_dtype == np.float64 else None
                if got is not None:
                    self.busy[j] = got[1]
                    return ("pinned", got[0], got[1]), Xr
                t = self.staging.stage([tau, Lu, Ld[None, :]], stream=self.side)
            self.busy[j] = t[3]
        return t, Xr

    def finish(self, ticket, dtype):
        if ticket[0] == "pinned":
            ticket[2].synchronize()
            return ticket[1]
        with torch.cuda.device(self.dev):
            return self.staging.collect(ticket, dtype)

    def close(self):
        with torch.cuda.device(self.dev):
            torch.cuda.synchronize()
            for pl in self.plans:
                pl.close()


def compute_TUD_batch(Xmin, Xmax, atmospheres, opts=options, reduce=None, devices=None, out_dtype=np.float64, **kwargs):
    """compute_TUD for MANY atmospheres on one spectral grid -- the reference's outer loop
    (Generate_LWIR_TUD.py:117-150: 199 atmospheres x `rt.compute_TUD(..., MFs_VAL=, Ts=, ...)`, fanned out over a
    multiprocessing.Pool(6) there) as device pipelines driven by ONE host process: atmosphere k's device-to-host copy
    runs on a side stream while the next atmosphere's kernels run on the compute stream.

    atmospheres: sequence of dicts with any of Ts, Ps, PLs, MFs_VAL (the per-atmosphere kwargs of the reference's caller);
    everything else (Zs, MFs_ID, DVOUT, Altitudes, theta_r, N_angle, returnOD, line_table) comes from opts / kwargs and is
    common to the batch. reduce = dict(dX=..., N=4, window="hanning") applies reduceResolution (:1327-1350) to tau, Lu and
    Ld on the device, as the reference's caller does right after each compute_TUD (Generate_LWIR_TUD.py:124-126): only the
    reduced spectra cross PCIe then.
    devices: GPU indices to spread the atmospheres over (the reference's Pool axis without a launcher): atmosphere k goes
    to devices[k % len(devices)]; the line table is uploaded once per distinct device; an index may repeat (independent
    pipelines sharing a device). None = the current device. Results come back in input order and are bit-identical to
    per-call compute_TUD whatever the devices.
    out_dtype: float64 (the reference's) or float32 (half the host work when full spectra are kept).
    Returns a list of (X, tau, Lu, Ld) with the shapes compute_TUD gives (X_out instead of X when reduce is set). Full
    float64 spectra are views of page-locked blocks while fewer than _hostio.PINNED_RESULT_CAP bytes are lent out (1 GiB:
    the first eight C3-sized results), ordinary pageable arrays beyond that and for every other output (float32 crosses
    PCIe into a reusable pinned ring, a host thread pool widens: host-memory bound, ~10 ms per 132 MB result)."""
    o = dict(opts)
    o.update(kwargs)
    if o["save"]:
        raise ValueError("compute_TUD_batch: save is a single-call option")
    if np.dtype(out_dtype) not in (np.dtype(np.float64), np.dtype(np.float32)):
        raise ValueError("compute_TUD_batch: out_dtype must be float64 or float32")
    Z = np.asarray(o["Zs"], dtype=np.float64)
    ID = np.asarray(o["MFs_ID"])
    f = lambda x: np.array([x]).ravel()
    Z_s = f(o["Altitudes"])
    mu_s = f(1.0 / np.cos(o["theta_r"]))
    if mu_s.size > engine.TUD_MAX_MU:
        raise ValueError("compute_TUD_batch takes at most %d slant paths" % engine.TUD_MAX_MU)
    engine.require_gpu()
    devs = [torch.cuda.current_device()] if devices is None else [int(d) for d in devices]
    if not devs or any(d < 0 or d >= torch.cuda.device_count() for d in devs):
        raise ValueError("compute_TUD_batch: devices=%r, visible GPUs: %d" % (devices, torch.cuda.device_count()))
    X_ = _cached_axis(Xmin, Xmax, o["DVOUT"])
    grid = engine.Grid(Xmin, Xmax, X_.size)
    with torch.cuda.device(devs[0]):
        tbl = _resolve_table(o.get("line_table"))
    nL = np.asarray(o["Ts"]).size
    theta = np.asarray(o["theta_r"], dtype=np.float64)
    pipes = [_TudPipeline(d, tbl, grid, Z, nL, Z_s, theta, int(o["N_angle"]), bool(o["returnOD"])) for d in devs]
    nZ, nMu = pipes[0].runs[0].shape
    pending = []  # (pipeline, ticket, X of the result) in input order
    results = []

    def finish(item):
        pipe, ticket, Xr = item
        tau_h, Lu_h, Ld_h = pipe.finish(ticket, np.dtype(out_dtype))
        tau_, Lu_ = _tud_shapes(tau_h, Lu_h, nZ, nMu)
        results.append((Xr, tau_, Lu_, Ld_h[0]))

    try:
        for k, atm in enumerate(atmospheres):
            a = dict(o)
            a.update(atm)
            pipe = pipes[k % len(pipes)]
            ticket, Xr = pipe.enqueue(a, ID, grid, float(X_[0]), reduce, np.dtype(out_dtype))
            pending.append((pipe, ticket, X_ if Xr is None else Xr))
            # a pipeline's staging ring has two slots: the atmosphere staged two rounds ago on it must be collected first
            while len(pending) > len(pipes):
                finish(pending.pop(0))
        while pending:
            finish(pending.pop(0))
    finally:
        for p_ in pipes:
            p_.close()
    return results


def compute_LWIR_apparent_radiance(X, emis, Ts, tau, La, Ld, dT=None, return_Ls=False):
    r"""L = tau [emis B(Ts + dT) + (1 - emis) Ld] + La for every combination, signature of :1017-1069.

    X (nX,), emis (nX,nE), Ts (nA,), tau/La/Ld (nX,nA), dT (nT,) optional ->
    L (nX,nE,nA) or (nX,nE,nA,nT) [, Ls]. Computed in float32 (the reference's own caller casts its
    inputs to float32 first, Compute_LWIR_Apparent_Radiance.py:9-20); NumPy in -> same dtype as `emis` out.
    """
    as_torch = any(_is_torch(v) for v in (X, emis, Ts, tau, La, Ld))
    dev = engine.device()
    Xd = _dev_f64(X, dev).flatten()
    em = _dev_f32(emis, dev)
    Tsd = _dev_f64(Ts, dev).flatten()
    nX, nA = Xd.numel(), Tsd.numel()
    td, Lad, Ldd = (_dev_f32(v, dev).reshape(nX, nA) for v in (tau, La, Ld))
    dTd = _dev_f64(np.asarray(dT).flatten() if not _is_torch(dT) else dT.flatten(), dev) if dT is not None else None
    L, Ls = engine.apparent_radiance(Xd, em.reshape(nX, -1), Tsd, td, Lad, Ldd, dTd, return_Ls)
    if dT is None:
        L = L[..., 0]
        Ls = Ls[..., 0] if Ls is not None else None
    if not as_torch:
        odt = np.asarray(emis).dtype if np.asarray(emis).dtype in (np.float32, np.float64) else np.float64
        L = L.cpu().numpy().astype(odt, copy=False)
        Ls = Ls.cpu().numpy().astype(odt, copy=False) if Ls is not None else None
    return (L, Ls) if return_Ls else L


# 128 MAKO band centres [um] -- instrument constants listed at :1092-1223
_MAKO_UM = np.array([
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


def _ils_apply(kind, X, Y, centre, sigma):
    """Shared device path of both ILS variants: Y (nX,) or (nX,nS) -> (nB,) or (nB,nS)."""
    as_torch = _is_torch(Y)
    dev = engine.device()
    Yd = _dev_f32(Y, dev)
    one_d = Yd.dim() == 1
    if one_d:
        Yd = Yd[:, None].contiguous()
    Xh = X.detach().cpu().numpy() if _is_torch(X) else np.asarray(X, dtype=np.float64)
    try:
        grid, Xd = engine.Grid.from_axis(Xh), None
    except NotImplementedError:
        grid, Xd = None, _dev_f64(Xh, dev)  # explicit (non-uniform) ascending axis
    out = engine.ils(kind, Yd, _dev_f64(centre, dev), _dev_f64(sigma, dev), X=Xd, grid=grid)
    if one_d:
        out = out[:, 0]
    if as_torch:
        return out
    odt = np.asarray(Y).dtype if np.asarray(Y).dtype in (np.float32, np.float64) else np.float64
    return out.cpu().numpy().astype(odt)


def ILS_MAKO(X, Y, resFactor=None, returnX=True, fwhm_sf=1.0, shift=0.0, scale=1.0):
    """MAKO instrument line shape (triangle), signature of :1072-1263.

    X (nX,) ascending wavenumbers, Y (nX,) or (nX,nS) -> X_out (nB,), Y_out (nB,) or (nB,nS);
    nB = 128 (or int(128*resFactor)) minus the bands outside the open interval (X.min, X.max) (:1233).
    A band whose triangle covers no grid point comes out NaN, as in the reference (quirk 13)."""
    Xh = X.detach().cpu().numpy() if _is_torch(X) else np.asarray(X, dtype=np.float64)
    X_out = _MAKO_UM.copy()
    if resFactor is not None:
        _x0 = np.linspace(0, 1, len(X_out))
        _x1 = np.linspace(0, 1, int(len(X_out) * resFactor))
        X_out = np.interp(_x1, _x0, X_out)
    X_out = np.sort(10000.0 / X_out)
    X_out = X_out[(X_out > Xh.min()) & (X_out < Xh.max())]
    sigma_out = fwhm_sf * np.abs(np.gradient(X_out)) * 1.6
    Y_out = _ils_apply(0, Xh, Y, scale * X_out + shift, sigma_out)
    if returnX:
        return X_out, Y_out
    return Y_out


# ---- post-processing of TUD products ----------------------------------------
def _uniform_axis(X, what):
    Xh = X.detach().cpu().numpy() if _is_torch(X) else np.asarray(X, dtype=np.float64)
    Xh = np.asarray(Xh, dtype=np.float64).ravel()
    if Xh.size < 2:
        raise ValueError(f"{what}: X needs at least two points")
    h = (Xh[-1] - Xh[0]) / (Xh.size - 1)
    if not np.allclose(np.diff(Xh), h, rtol=0, atol=1e-6 * abs(h)):
        raise NotImplementedError(f"{what}: only uniform spectral axes are supported")
    return Xh, float(h)


def smooth(x, window_len=11, window="hanning"):
    """Window smoothing, signature and early returns of :1266-1324 (1-D input only; reflect-padded, output of the
    same length; an even window_len leaves the result half a sample off centre, as in the reference)."""
    x = np.asarray(x)
    if x.ndim != 1:
        print("smooth only accepts 1 dimension arrays.")
        return x
    if x.size < window_len:
        print("Input vector needs to be bigger than window size.")
        return x
    if window_len < 3:
        return x
    if window not in engine._WINDOWS:
        print("Window is on of 'flat', 'hanning', 'hamming', 'bartlett', 'blackman'")
        return x
    engine.require_gpu()
    taps, c = engine.window_taps(window_len, window)
    Y = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64)[None, :], device="cuda")
    return engine.fir_reflect(Y, taps, c)[0].cpu().numpy()


def reduceResolution(X, Y, dX, N=4, window="hanning", X_out=None):
    """Smooth to resolution dX and resample, signature of :1327-1350: X (nX,) uniform, Y (nX,) or (nX, nC)
    -> (X_out, Y_out) when X_out is None, else Y_out; Y_out (nOut,) or (nOut, nC), float64.

    The reference smooths with window length round(dX / mean(diff(X))) (forward and reversed, averaged), then puts
    scipy's cubic interp1d through all smoothed samples. Here: one fp64 FIR kernel and one cardinal-spline kernel.
    Conscious divergences: (1) nPts = ceil(N*span/dX)+1 is evaluated with a 1e-9 guard, because N*span/dX is normally
    an exact integer and the reference's value flips between two counts with the rounding of its convolution;
    (2) np.int (:1329,1336) is int; (3) X_out must be ascending when it reaches within ceil(window/2) + 20 samples of an
    end of the axis, and must not leave the smoothed axis (interp1d's extrapolation is not reproduced). Output points near
    the ends -- where the reference's knots are the SMOOTHED, no longer uniform axis and its not-a-knot end condition acts --
    are evaluated by a local not-a-knot spline on those knots (rtx_cubic_end), so short windows work with the default X_out."""
    Xh, h = _uniform_axis(X, "reduceResolution")
    engine.require_gpu()
    Yt = Y if _is_torch(Y) else torch.as_tensor(np.asarray(Y, dtype=np.float64))
    one_d = Yt.dim() == 1
    Y2 = (Yt[None, :] if one_d else Yt.transpose(0, 1)).to("cuda").contiguous()
    if Y2.shape[1] != Xh.size:
        raise ValueError("reduceResolution: Y's first axis must match X")
    if Y2.dtype not in (torch.float32, torch.float64):
        Y2 = Y2.to(torch.float64)
    x_out, out = engine.reduce_resolution(Y2, float(Xh[0]), h, Xh.size, float(dX), N=N, window=window,
                                          x_out=None if X_out is None else np.asarray(X_out, dtype=np.float64).ravel())
    Y_out = out[0].cpu().numpy() if one_d else out.transpose(0, 1).contiguous().cpu().numpy()
    if X_out is None:
        return x_out, Y_out
    return Y_out


# ---- small host helpers of the reference module, kept for drop-in completeness -------------------------
def rs1D(y):
    """(flattened copy, original shape), as radiative_transfer.py:186-203."""
    y = np.array(y)
    return y.flatten(), y.shape


def rs2D(y):
    """(2-D view with the 2nd..Nth axes flattened, original shape); a scalar or 1-D input becomes one row
    (radiative_transfer.py:206-228)."""
    y = np.array(y)
    if y.ndim < 2:
        y = np.array([y]).flatten()[np.newaxis, :]
        return y, y.shape
    dims = y.shape
    return y.reshape((dims[0], int(np.prod(dims[1:])))), dims


def rsND(y, dims):
    """Back to the N-D shape (radiative_transfer.py:231-248)."""
    return y.reshape(dims)


def _no_lblrtm(name):
    def stub(*args, **kwargs):
        raise NotImplementedError(
            f"{name}: the LBLRTM plumbing of the reference (TAPE5 writer, binary runner, TAPE12 reader; "
            "radiative_transfer.py:395-789) is not part of this engine -- the LBLRTM executable and the AER line file are "
            "git-LFS stubs in the reference checkout. compute_OD / compute_TUD take a HITRAN-format line table instead "
            "(line_table=..., INTEGRATION.md section 2).")
    stub.__name__ = name
    return stub


write_tape5 = _no_lblrtm("write_tape5")
run_LBLRTM = _no_lblrtm("run_LBLRTM")
read_tape12 = _no_lblrtm("read_tape12")
