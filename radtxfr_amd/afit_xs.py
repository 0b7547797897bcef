"""AFIT_XS cross-section files: the binary format written by the reference's cross-section generator
(misc/RT_gen_AbsXS_files.py:45-83) and a batched generator for its temperature x pressure loop (:86-92).

File layout (little endian), as the reference's `AFIT_XS_write` produces it (its comments quote other sizes):
    2 bytes   b"v1"
    48 bytes  six float64: X.min(), X.max(), X.size, molecule ID, T [K], P [Pa]
    128 bytes database description, NUL padded
    8*n bytes the cross section as float64
Default file name: XS-{ID:02d}-{T:04d}K-{P:06d}Pa.bin (:73).

The reference computes one (T, p) state per hapi call; here all states of the grid go through ONE prologue + line-sum
launch as "layers" (rtx_line_prep + rtx_voigt_sum), 128 states at a time."""
import struct

import numpy as np
import torch

from . import _hostio, _lib, engine
from . import hapi as _hapi

def _touch_pages(a):
    """Write one element per 4 KiB page of a fresh float64 slice (the values are overwritten by the copy that follows)."""
    a[::512] = 0.0
    a[-1:] = 0.0


_STAGE = {}  # one reusable pinned staging block for cross_section_grid (grow-only)
_HEADER = struct.Struct("<2s6d128s")


def AFIT_XS_write(X, Y, T, P, ID, fnDB, File=[]):
    """Save a cross section in the AFIT_XS format; same arguments and return value as RT_gen_AbsXS_files.py:45-83."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y)
    if len(File) == 0:
        File = "XS-{0:02d}-{1:04d}K-{2:06d}Pa.bin".format(int(ID), int(T), int(P))
    db = str(fnDB).encode()[:128]
    with open(File, "wb") as f:
        f.write(_HEADER.pack(b"v1", float(X.min()), float(X.max()), float(X.size), float(ID), float(T), float(P), db))
        f.write(Y.astype("<f8").tobytes())
    return File


def AFIT_XS_read(File):
    """Inverse of AFIT_XS_write: dict(version, Xmin, Xmax, n, ID, T, P, db, X, Y)."""
    with open(File, "rb") as f:
        raw = f.read()
    ver, xmin, xmax, n, mid, T, P, db = _HEADER.unpack_from(raw, 0)
    n = int(n)
    Y = np.frombuffer(raw, dtype="<f8", count=n, offset=_HEADER.size).copy()
    return {"version": ver.decode(), "Xmin": xmin, "Xmax": xmax, "n": n, "ID": int(mid), "T": T, "P": P,
            "db": db.rstrip(b"\0").decode(), "X": np.linspace(xmin, xmax, n), "Y": Y}


def cross_section_grid(SourceTables, T, P_atm, X, WavenumberWingHW=50.0, WavenumberWing=0.0, IntensityThreshold=0.0,
                       GammaL="gamma_air", Components=None):
    """HITRAN-unit Voigt cross sections [cm^2/molecule] of one table for every (T, p) pair of the grids: the body of the
    reference's double loop (RT_gen_AbsXS_files.py:88-90: absorptionCoefficient_SDVoigt, i.e. the Voigt line-sum for tables
    without speed-dependence columns and rtx_sdvoigt_sum for tables with them) as one batched launch per 128 states. X must be uniform. Returns xs[nT][nP][nX] float64 (host)."""
    T = np.atleast_1d(np.asarray(T, dtype=np.float64))
    P = np.atleast_1d(np.asarray(P_atm, dtype=np.float64))
    X = np.asarray(X, dtype=np.float64)
    engine.require_gpu()
    tbl = _hapi._device_table(_hapi.listOfTuples(SourceTables))
    grid = engine.Grid.from_axis(X)
    comps = [p for p in tbl.species if p != (0, 0)] if Components is None else [(int(c[0]), int(c[1])) + tuple(c[2:3]) for c in Components]
    w = np.zeros(len(tbl.species))
    for s, mi in enumerate(tbl.species):
        for c in comps:
            if (c[0], c[1]) == mi:
                nat = _hapi.abundance(*mi)
                w[s] = (c[2] if len(c) >= 3 else nat) / nat
    states = [(t, p) for t in T for p in P]
    smax = float(np.max(tbl.cols["sw"])) * float(np.max(w)) if tbl.n and np.max(w) > 0 else 1.0
    scale = 2.0 ** (-np.floor(np.log2(smax))) if smax > 0 and np.isfinite(smax) else 1.0
    dil = {"air": 1.0} if GammaL.lower() == "gamma_air" else {"self": 1.0}
    out = np.empty((len(states), X.size))
    # states per launch: the per-(line, state) records (80-128 B each) and, with the caller's 350-half-width wings, the
    # partial tiles of the line-sum's part list (every tile then has > 256 candidate lines) must fit comfortably
    per = max(1, min(128, int(2.0e9 / (128.0 * max(tbl.n, 1)))))
    s0 = 0
    while s0 < len(states):
        chunk = states[s0:s0 + per]
        Tk = np.array([c[0] for c in chunk])
        pk = np.array([c[1] for c in chunk])
        if tbl.n:
            dev = torch.empty((len(chunk), grid.n), dtype=torch.float64, device=engine.device())
            try:
                engine.voigt_sum(tbl, grid, Tk, pk, np.tile(w[:, None], (1, len(chunk))), out_f64=dev, dil_air=dil.get("air", 0.0),
                                 dil_self=dil.get("self", 0.0), omega_wing=WavenumberWing, omega_wing_hw=WavenumberWingHW,
                                 intensity_threshold=IntensityThreshold, scale=scale, profile=3 if tbl.has_sd else 0)
            except _lib.RtxError as e:
                if "work list" in str(e) and per > 1:  # too many partial tiles for one launch: fewer states at a time
                    per = max(1, per // 4)
                    continue
                raise
            # while the kernels run: the host thread pool touches one word per page of this chunk's rows of `out` -- fresh
            # memory whose page faults (8-90 ms for 86 MB, box and moment dependent) would otherwise sit behind the copy
            rows = out[s0:s0 + len(chunk)]
            touch = [_hostio._threads().submit(_touch_pages, rows[r, c0:c0 + _hostio._CHUNK])
                     for r in range(len(chunk)) for c0 in range(0, grid.n, _hostio._CHUNK)]
            # device -> one reusable pinned block (a pageable hipMemcpy runs at a few GB/s) -> the result rows, copied by
            # the host thread pool
            need = len(chunk) * grid.n
            pinned = _STAGE.get("buf")
            if pinned is None or pinned.numel() < need:  # kept between calls: page-locking a block of this size costs ~70 ms
                pinned = _STAGE["buf"] = torch.empty((need,), dtype=torch.float64, pin_memory=True)
            pinned = pinned[:need].view(len(chunk), grid.n)
            pinned[:len(chunk)].copy_(dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            src = pinned.numpy()
            for f in touch:
                f.result()
            futs = [_hostio._threads().submit(np.copyto, out[s0 + r, c0:c0 + _hostio._CHUNK], src[r, c0:c0 + _hostio._CHUNK])
                    for r in range(len(chunk)) for c0 in range(0, grid.n, _hostio._CHUNK)]
            for f in futs:
                f.result()
        else:
            out[s0:s0 + len(chunk)] = 0.0
        s0 += len(chunk)
    return out.reshape(T.size, P.size, X.size)


def generate_xs_files(SourceTables, ID, T, P_atm, X, descr, WavenumberWingHW=50.0, directory="."):
    """The reference's generator loop (RT_gen_AbsXS_files.py:86-92) for one molecule table: one AFIT_XS file per (T, p);
    returns the file names in the reference's order (T outer, p inner). P is written in Pa (101325*p, :91)."""
    import os

    T = np.atleast_1d(np.asarray(T, dtype=np.float64))
    P = np.atleast_1d(np.asarray(P_atm, dtype=np.float64))
    xs = cross_section_grid(SourceTables, T, P, X, WavenumberWingHW=WavenumberWingHW)
    names = []
    for it, t in enumerate(T):
        for ip, p in enumerate(P):
            fn = "XS-{0:02d}-{1:04d}K-{2:06d}Pa.bin".format(int(ID), int(t), int(101325 * p))
            names.append(AFIT_XS_write(X, xs[it, ip], t, 101325 * p, ID, descr, File=os.path.join(directory, fn)))
    return names
