"""At-sensor band radiances on device tensors: the C4 pipeline of SURVEY.md 8d
(emissivity knots -> monochromatic grid -> compute_LWIR_apparent_radiance -> ILS_MAKO) without ever
leaving the GPU. The reference runs these steps as separate scripts on band-averaged TUDs
(Compute_LWIR_Apparent_Radiance.py:25 on the output of Generate_LWIR_TUD_MAKO.py:34-36) because its
ILS needs an (nS,nX,nB) temporary (1.4 TB at 2000 spectra); here the monochromatic arrays fit in HBM
(560 k wavenumbers x 2000 spectra x 4 B = 4.5 GB) and the order "radiance first, ILS second" is exact.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _lib, engine
from .radiative_transfer import _MAKO_UM


def interp_knots(grid, Xk, F):
    """np.interp(grid, Xk, F[:, s]) for every column: F [nk][nS] float32 device -> [grid.n][nS] float32."""
    lib = _lib.load()
    assert F.dtype == torch.float32 and F.is_cuda and F.dim() == 2
    F = F.contiguous()
    Xk_d = torch.as_tensor(np.asarray(Xk, dtype=np.float64), device=F.device).contiguous()
    out = torch.empty((grid.n, F.shape[1]), dtype=torch.float32, device=F.device)
    _lib.check(lib.rtx_interp_knots(grid.byref(), None, grid.n, C.c_void_p(Xk_d.data_ptr()), Xk_d.numel(),
                                    C.c_void_p(F.data_ptr()), F.shape[1], C.c_void_p(out.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def mako_bands(x_min, x_max, resFactor=None, fwhm_sf=1.0, shift=0.0, scale=1.0):
    """Band axis, centres and triangle widths of rt.ILS_MAKO (radiative_transfer.py:1226-1241)."""
    X_out = _MAKO_UM.copy()
    if resFactor is not None:
        X_out = np.interp(np.linspace(0, 1, int(len(X_out) * resFactor)), np.linspace(0, 1, len(X_out)), X_out)
    X_out = np.sort(10000.0 / X_out)
    X_out = X_out[(X_out > x_min) & (X_out < x_max)]
    sigma = fwhm_sf * np.abs(np.gradient(X_out)) * 1.6
    return X_out, scale * X_out + shift, sigma


_CUBE_PLANS = {}  # spectra-independent set-up of hsi_cube (band list, knots and node tables on the device), a few entries
_CUBE_PLANS_MAX = 8
_CUBE_PLANS_LOCK = threading.Lock()
_FUSED_PLANS = {}  # band list and knot axis of band_radiance_fused on the device (a few entries, see _cube_plan)


def _fused_plan(grid, Xk, resFactor, kind, dev):
    Xk = np.ascontiguousarray(Xk, dtype=np.float64)
    key = (grid.x_at(0), grid.step, grid.n, resFactor, int(kind), str(dev), Xk.tobytes())
    with _CUBE_PLANS_LOCK:
        plan = _FUSED_PLANS.get(key)
    if plan is None:
        if kind == 0:
            X_out, centre, sigma = mako_bands(grid.x_at(0), grid.x_at(grid.n - 1), resFactor)
        else:  # Gaussian variant: no clipping, sigma = |gradient| (ILS_MAKO.py:19-21)
            X_out = np.sort(10000.0 / _MAKO_UM)
            centre, sigma = X_out, np.abs(np.gradient(X_out))
        plan = {"X_out": np.array(X_out, dtype=np.float64), "Xk_d": torch.as_tensor(Xk, device=dev),
                "c_d": torch.as_tensor(np.ascontiguousarray(centre, dtype=np.float64), device=dev),
                "s_d": torch.as_tensor(np.ascontiguousarray(sigma, dtype=np.float64), device=dev)}
        with _CUBE_PLANS_LOCK:
            if len(_FUSED_PLANS) >= _CUBE_PLANS_MAX:
                _FUSED_PLANS.pop(next(iter(_FUSED_PLANS)))
            _FUSED_PLANS[key] = plan
    return plan


def band_radiance_fused(grid, tau, La, Ld, Xk, emis_knots, Ts, resFactor=None, kind=0):
    """Same result as band_radiance() without any [nX][nE] array: one monochromatic pass
    (rtx_band_moments) + a [nB x nk] x [nk x nE] contraction over each band's ~10-30 knots (rtx_band_mix).
    Returns (X_out [nB] NumPy, L [nB][nE] float32 device)."""
    lib = _lib.load()
    dev = tau.device
    plan = _fused_plan(grid, Xk, resFactor, kind, dev)
    X_out = plan["X_out"].copy()
    nB, nk, nE = X_out.size, len(Xk), emis_knots.shape[1]
    assert emis_knots.dtype == torch.float32 and emis_knots.shape[0] == nk
    emis_knots = emis_knots.contiguous()
    N = torch.empty(nB, dtype=torch.float32, device=dev)
    Cb = torch.empty(nB, dtype=torch.float32, device=dev)
    M = torch.empty((nB, nk), dtype=torch.float32, device=dev)
    jr = torch.empty((nB, 2), dtype=torch.int32, device=dev)
    out = torch.empty((nB, nE), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rtx_band_moments(int(kind), grid.byref(), p(tau), p(La), p(Ld), float(Ts), p(plan["Xk_d"]), nk, nB, p(plan["c_d"]),
                                    p(plan["s_d"]), p(N), p(Cb), p(M), p(jr), st))
    _lib.check(lib.rtx_band_mix(p(N), p(Cb), p(M), p(jr), nB, nk, p(emis_knots), nE, p(out), st))
    return X_out, out


def chebyshev_lagrange(Q):
    """Chebyshev nodes s_q on (-1,1) and the monomial coefficients coef[q][d] of their Lagrange basis."""
    s = np.cos((2 * np.arange(Q) + 1) * np.pi / (2 * Q))
    coef = np.zeros((Q, Q))
    for q in range(Q):
        others = np.delete(s, q)
        poly = np.poly(others) / np.prod(s[q] - others)  # highest power first
        coef[q] = poly[::-1]
    return s, coef




def _cube_plan(grid, Xk, resFactor, band_slice, Q, bands, dev):
    """Everything hsi_cube needs that does not depend on the spectra or the scene: the band list of rt.ILS_MAKO, its centres
    and widths and the knot axis as device tensors, the Chebyshev nodes and Lagrange coefficients. A scene generator calls
    hsi_cube once per atmosphere / scene with the same axis, knots and bands; building these took as long as the kernels."""
    Xk = np.ascontiguousarray(Xk, dtype=np.float64)
    key = (grid.x_at(0), grid.step, grid.n, resFactor, band_slice, Q, str(dev), Xk.tobytes(),
           None if bands is None else tuple(np.ascontiguousarray(v, dtype=np.float64).tobytes() for v in bands))
    with _CUBE_PLANS_LOCK:
        plan = _CUBE_PLANS.get(key)
    if plan is None:
        X_out, centre, sigma = bands if bands is not None else mako_bands(grid.x_at(0), grid.x_at(grid.n - 1), resFactor)
        if band_slice is not None:
            X_out, centre, sigma = (v[band_slice[0]:band_slice[1]] for v in (X_out, centre, sigma))
        s_nodes, coef = chebyshev_lagrange(Q)
        plan = {"X_out": np.array(X_out, dtype=np.float64), "coef32": np.ascontiguousarray(coef, dtype=np.float32),
                "sn32": np.ascontiguousarray(s_nodes, dtype=np.float32), "Xk_d": torch.as_tensor(Xk, device=dev),
                "c_d": torch.as_tensor(np.ascontiguousarray(centre, dtype=np.float64), device=dev),
                "s_d": torch.as_tensor(np.ascontiguousarray(sigma, dtype=np.float64), device=dev)}
        with _CUBE_PLANS_LOCK:
            if len(_CUBE_PLANS) >= _CUBE_PLANS_MAX:
                _CUBE_PLANS.pop(next(iter(_CUBE_PLANS)))
            _CUBE_PLANS[key] = plan
    return plan


def hsi_cube(grid, tau, La, Ld, Xk, endmembers, kidx, frac, Tpix, resFactor=2, band_slice=None, Q=4, bands=None):
    """Config C5: band radiances of an HSI cube whose pixels each have an emissivity mixture and a surface
    temperature of their own (LWIR_HSI_Generator.py:151-167), from monochromatic tau/La/Ld through the
    triangle ILS (rt.ILS_MAKO with resFactor).

    endmembers [nk][nEnd] float32 device (knot spectra), kidx [nPix][nMix] int32, frac [nPix][nMix] float32,
    Tpix [nPix] float64 -- device tensors. band_slice: (b0, b1) to compute only a band-aligned shard.
    bands = (X_out, centre, sigma): explicit band list (mako_bands() of the FULL spectral axis) when `grid` is only the
    wavenumber shard under those bands (dist.hsi_cube_from_atmosphere); default: the bands inside `grid`.
    Returns (X_out [nB] NumPy, cube [nB][nPix] float32 device). Three launches: rtx_band_basis_moments (the one pass over
    the monochromatic arrays), rtx_band_mix_stacked (its Q + 1 moment arrays x the endmember knots), rtx_pixel_cube."""
    lib = _lib.load()
    dev = tau.device
    plan = _cube_plan(grid, Xk, resFactor, None if band_slice is None else (int(band_slice[0]), int(band_slice[1])), int(Q), bands, dev)
    X_out = plan["X_out"].copy()
    nB, nk, nEnd = X_out.size, len(Xk), endmembers.shape[1]
    nPix, nMix = kidx.shape
    assert endmembers.dtype == torch.float32 and endmembers.shape[0] == nk
    endmembers = endmembers.contiguous()
    assert kidx.dtype == torch.int32 and frac.dtype == torch.float32 and Tpix.dtype == torch.float64
    f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    cube = f32(nB, nPix)
    if nB == 0:
        return X_out, cube
    N, Cb, M = f32(nB), f32(nB), f32(Q + 1, nB, nk)  # M[0..Q-1]: the Planck-node basis moments, M[Q]: the Ld moment
    jr = torch.empty((nB, 2), dtype=torch.int32, device=dev)
    tab = f32(nEnd, Q + 1, nB)
    p = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rtx_band_basis_moments(0, grid.byref(), p(tau), p(La), p(Ld), p(plan["Xk_d"]), nk, nB, p(plan["c_d"]), p(plan["s_d"]), Q,
                                          plan["coef32"].ctypes.data_as(C.c_void_p), 1.0, p(N), p(Cb), p(M[Q]), p(M), p(jr), st))
    _lib.check(lib.rtx_band_mix_stacked(p(M), p(jr), nB, Q + 1, nk, p(endmembers), nEnd, p(tab), st))
    _lib.check(lib.rtx_pixel_cube(nB, Q, p(plan["c_d"]), p(plan["s_d"]), 1.0, plan["sn32"].ctypes.data_as(C.c_void_p), p(N), p(Cb), p(tab),
                                  nEnd, nPix, nMix, p(kidx.contiguous()), p(frac.contiguous()), p(Tpix.contiguous()), p(cube), st))
    return X_out, cube


def band_radiance(grid, tau, La, Ld, Xk, emis_knots, Ts, resFactor=None, keep_hires=False):
    """C4: L_b,k = ILS_MAKO( tau*(eps_k*B(Ts) + (1-eps_k)*Ld) + La ) for every emissivity column k.

    grid: engine.Grid of the monochromatic axis (e.g. the MAKO span of the C3 grid);
    tau, La, Ld: [grid.n] float32 device tensors; Xk [nk], emis_knots [nk][nE] float32 device; Ts scalar [K].
    Returns (X_out [nB] NumPy, L [nB][nE] float32 device). Three streaming kernels:
    rtx_interp_knots -> rtx_apparent_radiance -> rtx_ils."""
    dev = tau.device
    em = interp_knots(grid, Xk, emis_knots)  # [nX][nE]
    X_d = torch.as_tensor(grid.axis(), device=dev)
    Ts_d = torch.as_tensor(np.atleast_1d(np.asarray(Ts, dtype=np.float64)), device=dev)
    col = lambda v: v.reshape(-1, 1).contiguous()
    L, _ = engine.apparent_radiance(X_d, em, Ts_d, col(tau), col(La), col(Ld))
    L = L.reshape(grid.n, -1)  # [nX][nE] (nA = nT = 1)
    X_out, centre, sigma = mako_bands(grid.axis()[0], grid.axis()[-1], resFactor)
    out = engine.ils(0, L, torch.as_tensor(centre, device=dev), torch.as_tensor(sigma, device=dev), grid=grid)
    return (X_out, out, L) if keep_hires else (X_out, out)
