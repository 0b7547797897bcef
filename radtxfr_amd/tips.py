"""TIPS-2011 total internal partition sums and the isotopologue table (host side, fp64).

Mirrors the small per-species part of the reference's hapi chain that stays on the host:
PYTIPS / BD_TIPS_2011_PYTHON (misc/hapi.py:9568-9582, 10030) with its 3-/4-point Lagrange
interpolation AtoB (:5311-5388) on the nodes Tdat = 60:25:3010 K (:5401-5413), and the
ISO abundance / molar-mass table (:3372-3496, 5088-5124). The tables themselves are data
(radtxfr_amd/data/tips2011.npz, produced by oracle/make_tables.py).

Per call this is n_species * n_layers interpolations -- the per-LINE work (which the reference
repeats for every line, misc/hapi.py:11069-11070) runs on the GPU in rtx_line_prep.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "tips2011.npz")
_T = None


def _tab():
    global _T
    if _T is None:
        d = np.load(_DATA, allow_pickle=False)
        _T = {
            "tdat": d["tdat"].astype(np.float64),
            "q": {tuple(k): d["q"][r].astype(np.float64) for r, k in enumerate(d["mi"].tolist())},
            "abun": {tuple(k): float(v) for k, v in zip(d["iso_mi"].tolist(), d["iso_abun"])},
            "mass": {tuple(k): float(v) for k, v in zip(d["iso_mi"].tolist(), d["iso_mass"])},
        }
    return _T


def _lagrange(aa, A, B, idx):
    """Lagrange polynomial through nodes idx (3 or 4 of them), evaluated factor by factor in the
    same order as the reference so results agree to the last bits."""
    bb = 0.0
    for j in idx:
        num = 1.0
        den = 1.0
        for m in idx:
            if m != j:
                num = num * (aa - A[m])
                d = A[j] - A[m]
                den = den * (d if d != 0.0 else 0.0001)
        bb = bb + (num / den) * B[j]
    return bb


def partition_sum(M, I, T):
    """Q(T) for isotopologue (M,I); raises outside 70..3000 K like the reference (:9571)."""
    T = float(T)
    if T < 70.0 or T > 3000.0:
        raise Exception("TIPS: T must be between 70K and 3000K.")
    t = _tab()
    key = (int(M), int(I))
    if key not in t["q"] or not np.isfinite(t["q"][key][0]):
        raise Exception("TIPS: no data for M,I = %d,%d." % key)
    A, B = t["tdat"], t["q"][key]
    npt = A.size
    # first 1-based I >= 2 with A[I-1] >= T  (:5320-5321)
    I1 = int(np.searchsorted(A[1:], T, side="left")) + 2
    if I1 < 3 or I1 == npt:
        J = (3 if I1 < 3 else npt) - 1
        return _lagrange(T, A, B, (J - 2, J - 1, J))
    J = I1 - 1
    return _lagrange(T, A, B, (J - 2, J - 1, J, J + 1))


PYTIPS = partition_sum


class PartitionPlan:
    """partition_sums for a FIXED species list: the table rows are stacked once and the Lagrange denominators -- products
    of differences of table nodes, which are exact multiples of the uniform 25 K spacing -- are constants, so a call is
    ~25 NumPy operations on [4][nT] / [nS][nT] arrays. Same operations in the same order as the scalar routine for every
    temperature inside the table's 4-point range: bit-identical (tests/test_host.py); other temperatures take
    partition_sums()."""

    def __init__(self, species):
        t = _tab()
        self.species = [(int(m), int(i)) for m, i in species]
        self.A = t["tdat"]
        rows = []
        for key in self.species:
            if key not in t["q"] or not np.isfinite(t["q"][key][0]):
                raise Exception("TIPS: no data for M,I = %d,%d." % key)
            rows.append(t["q"][key])
        self.B = np.stack(rows) if rows else np.zeros((0, self.A.size))
        A = self.A
        h = np.diff(A)
        self.uniform = bool(np.all(h == h[0]))
        if self.uniform:
            a = A[:4]
            self.den = [float(np.prod([1.0] + [a[j] - a[m] for m in range(4) if m != j])) for j in range(4)]
            # the scalar routine multiplies den factor by factor starting from 1.0: exact here, every factor is a multiple of h
            for j in range(4):
                d = 1.0
                for m in range(4):
                    if m != j:
                        d = d * (a[j] - a[m])
                assert d == self.den[j]
        self._k4 = np.arange(4)[:, None]

    def __call__(self, T):
        T = np.atleast_1d(np.asarray(T, dtype=np.float64))
        A, npt = self.A, self.A.size
        if T.size and (T.min() < 70.0 or T.max() > 3000.0):
            raise Exception("TIPS: T must be between 70K and 3000K.")
        I1 = np.searchsorted(A[1:], T, side="left") + 2
        if not self.uniform or I1.min() < 3 or I1.max() >= npt:
            return partition_sums(self.species, T)
        idx = (I1 - 3)[None, :] + self._k4            # nodes J-2 .. J+1 with J = I1 - 1
        d0, d1, d2, d3 = T - A[idx]                   # [4][nT] -> rows
        den = self.den
        Bn = self.B[:, idx]                           # [nS][4][nT]
        p01 = d0 * d1
        out = (((d1 * d2) * d3) / den[0]) * Bn[:, 0] + 0.0
        out = out + (((d0 * d2) * d3) / den[1]) * Bn[:, 1]
        out = out + ((p01 * d3) / den[2]) * Bn[:, 2]
        out = out + ((p01 * d2) / den[3]) * Bn[:, 3]
        return out


def partition_sums(species, T):
    """Q[s][k] = partition_sum(*species[s], T[k]) for all species and temperatures at once: the same 3-/4-point
    Lagrange arithmetic in the same order (bit-identical to the scalar routine, tests/test_host.py), as a few dozen
    NumPy operations on [nS][nT] arrays instead of nS*nT Python calls -- this runs once per atmosphere on the host
    inside the timed step of bench.py."""
    T = np.atleast_1d(np.asarray(T, dtype=np.float64))
    if T.size and (T.min() < 70.0 or T.max() > 3000.0):
        raise Exception("TIPS: T must be between 70K and 3000K.")
    t = _tab()
    A = t["tdat"]
    npt = A.size
    rows = []
    for (M, I) in species:
        key = (int(M), int(I))
        if key not in t["q"] or not np.isfinite(t["q"][key][0]):
            raise Exception("TIPS: no data for M,I = %d,%d." % key)
        rows.append(t["q"][key])
    B = np.stack(rows) if rows else np.zeros((0, npt))
    I1 = np.searchsorted(A[1:], T, side="left") + 2
    three = (I1 < 3) | (I1 == npt)
    J = np.where(I1 < 3, 2, np.where(I1 == npt, npt - 1, I1 - 1))
    idx = np.stack([J - 2, J - 1, J, np.minimum(J + 1, npt - 1)])      # [4][nT]; the 4th node unused where `three`
    An = A[idx]                                                         # [4][nT]
    Bn = B[:, idx]                                                      # [nS][4][nT]
    out = np.zeros((B.shape[0], T.size))
    if not three.any():  # every temperature inside the table (the normal case): plain 4-point form, distinct nodes
        dT = [T - An[m] for m in range(4)]
        for j in range(4):
            num = 1.0
            den = 1.0
            for m in range(4):
                if m != j:
                    num = num * dT[m]
                    den = den * (An[j] - An[m])
            out = out + (num / den) * Bn[:, j, :]
        return out
    for j in range(4):
        num = np.ones(T.size)
        den = np.ones(T.size)
        for m in range(4):
            if m == j:
                continue
            use = ~three if m == 3 else np.ones(T.size, dtype=bool)
            d = An[j] - An[m]
            d = np.where(d != 0.0, d, 0.0001)
            num = np.where(use, num * (T - An[m]), num)
            den = np.where(use, den * d, den)
        term = (num / den) * Bn[:, j, :]
        out = out + (np.where(three, 0.0, term) if j == 3 else term)
    return out


def abundance(M, I):
    """Natural abundance, misc/hapi.py:5088-5103."""
    try:
        return _tab()["abun"][(int(M), int(I))]
    except KeyError:
        raise Exception("cannot find component M,I = %d,%d." % (M, I))


def molecularMass(M, I):
    """Molar mass [g/mol], misc/hapi.py:5109-5124."""
    try:
        return _tab()["mass"][(int(M), int(I))]
    except KeyError:
        raise Exception("cannot find component M,I = %d,%d." % (M, I))


def known_isotopologues():
    return sorted(_tab()["abun"].keys())
