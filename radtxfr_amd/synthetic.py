"""Synthetic HITRAN-format inputs.

No HITRAN line data and no AER TAPE3 exist offline, so parity and benchmarks use
synthetic line tables with the column set and the `.par` field precision of the
reference's line-table type (misc/hapi.py:468-559: nu %12.6f, sw %10.3E,
gamma_air %5.4f, gamma_self %5.3f, elower %10.4f, n_air %4.2f, delta_air %8.6f).
The same table feeds the oracle and the HIP engine.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# seeds fixed by SURVEY.md section 8d
SEED_C2 = 20261004
SEED_C3 = 20261005
SEED_C4 = 20261006
SEED_C5 = 42


def synth_line_table(seed, n_lines, nu_lo, nu_hi):
    """Columns as numpy arrays, sorted by nu; values rounded to .par precision."""
    rng = np.random.default_rng(seed)
    nu = np.sort(rng.uniform(nu_lo, nu_hi, n_lines))
    molec = rng.integers(1, 3, n_lines)  # 1 = H2O, 2 = CO2, p = 1/2 each
    iso = np.where(rng.random(n_lines) < 0.9, 1, 2)
    sw = 10.0 ** rng.uniform(-27.0, -19.0, n_lines)
    elower = rng.uniform(0.0, 3000.0, n_lines)
    gamma_air = rng.uniform(0.03, 0.11, n_lines)
    gamma_self = rng.uniform(0.1, 0.5, n_lines)
    n_air = rng.uniform(0.4, 0.8, n_lines)
    delta_air = rng.uniform(-0.01, 0.002, n_lines)
    nu = np.sort(np.round(nu, 6))
    sw = np.array([float("%10.3E" % v) for v in sw])
    return {
        "molec_id": molec.astype(np.int64),
        "local_iso_id": iso.astype(np.int64),
        "nu": nu,
        "sw": sw,
        "elower": np.round(elower, 4),
        "gamma_air": np.round(gamma_air, 4),
        "gamma_self": np.round(gamma_self, 3),
        "n_air": np.round(n_air, 2),
        "delta_air": np.round(delta_air, 6),
    }


def synth_clustered_table(seed, n_lines, nu_lo, nu_hi, n_heads=12, head_lines=3000, head_width=0.5):
    """A line table shaped like real line lists rather than like uniform noise (H2O / CO2 lists cluster: Q-branches with
    hundreds of lines per cm^-1, repeated centres, combs, empty windows between bands). Same columns, distributions of
    the line parameters and `.par` rounding as synth_line_table; what differs is WHERE the lines sit:
      * n_heads band heads: head_lines lines inside head_width cm^-1, their centres drawn from ~head_lines/4 distinct
        values (so identical nu occur, as blended lines do), strengths spanning the whole 1e-30 ... 1e-19 in one tile;
      * a P/R comb beside every head: spacing ~ 2B(1 -/+ J/J_max) shrinking towards the head, B ~ 0.4 ... 1.9 cm^-1;
      * the remainder uniform over 40 % of the span only, so whole stretches hold no line or one isolated line.
    Sorted by nu. The reference's per-line loop (misc/hapi.py:11050) has no notion of tiles; this is the table that makes
    a tiled engine show whether a hot tile serialises a launch."""
    rng = np.random.default_rng(seed)
    span = nu_hi - nu_lo
    heads = np.sort(rng.uniform(nu_lo + 0.05 * span, nu_hi - 0.05 * span, n_heads))
    parts = []
    for h in heads:
        distinct = h + rng.uniform(0.0, head_width, max(1, head_lines // 4))
        parts.append(rng.choice(distinct, head_lines))
    n_left = n_lines - n_heads * head_lines
    assert n_left > 0, "n_lines too small for the band heads"
    n_comb = n_left // 2
    per = n_comb // (2 * n_heads)
    for h in heads:
        B = rng.uniform(0.4, 1.9)
        J = np.arange(1, per + 1)
        d = np.cumsum(2.0 * B * np.maximum(0.02, 1.0 - J / (1.15 * per)) * 40.0 / per)  # ~40 cm^-1 per branch
        parts.append(h - d)
        parts.append(h + head_width + d)
    n_iso = n_lines - sum(p.size for p in parts)
    # isolated / sparse remainder: uniform over random sub-intervals covering 40 % of the span
    edges = np.sort(rng.uniform(nu_lo, nu_hi, 40))
    keep = rng.random(edges.size - 1) < 0.4
    keep[0] = True
    seg_lo, seg_hi = edges[:-1][keep], edges[1:][keep]
    w = (seg_hi - seg_lo) / (seg_hi - seg_lo).sum()
    # one isolated line in the middle of every dropped stretch (those away from a comb have no neighbour for tens of cm^-1)
    mid = 0.5 * (edges[:-1][~keep] + edges[1:][~keep])
    parts.append(mid)
    n_iso -= mid.size
    seg = rng.choice(seg_lo.size, n_iso, p=w)
    parts.append(seg_lo[seg] + rng.random(n_iso) * (seg_hi - seg_lo)[seg])
    nu = np.clip(np.concatenate(parts), nu_lo, nu_hi)
    n = nu.size
    molec = rng.integers(1, 3, n)
    iso = np.where(rng.random(n) < 0.9, 1, 2)
    sw = 10.0 ** rng.uniform(-30.0, -19.0, n)
    elower = rng.uniform(0.0, 3000.0, n)
    gamma_air = rng.uniform(0.03, 0.11, n)
    gamma_self = rng.uniform(0.1, 0.5, n)
    n_air = rng.uniform(0.4, 0.8, n)
    delta_air = rng.uniform(-0.01, 0.002, n)
    order = np.argsort(np.round(nu, 6), kind="stable")
    f = lambda v: v[order]
    return {
        "molec_id": f(molec.astype(np.int64)),
        "local_iso_id": f(iso.astype(np.int64)),
        "nu": f(np.round(nu, 6)),
        "sw": f(np.array([float("%10.3E" % v) for v in sw])),
        "elower": f(np.round(elower, 4)),
        "gamma_air": f(np.round(gamma_air, 4)),
        "gamma_self": f(np.round(gamma_self, 3)),
        "n_air": f(np.round(n_air, 2)),
        "delta_air": f(np.round(delta_air, 6)),
    }


def subset_table(tbl, lo, hi):
    """Rows with lo <= nu <= hi (a line farther than its wing cutoff cannot contribute)."""
    m = (tbl["nu"] >= lo) & (tbl["nu"] <= hi)
    return {k: v[m] for k, v in tbl.items()}


def load_standard_atmosphere():
    """66-row 1976 US-Std table: N,Z0,Z1,PL,P,T,H2O,CO2,O3,N2O,CO,CH4,O2,N2,Ar
    (copy of the reference's StandardAtmosphere.csv input fixture)."""
    return np.loadtxt(os.path.join(_DATA, "StandardAtmosphere.csv"), delimiter=",", skiprows=1)


def c3_atmosphere(n_layers=32):
    """Rows 1..n_layers of the CSV: the '32-layer' column of configs C3-C5 (0-9.5 km)."""
    A = load_standard_atmosphere()[:n_layers]
    return {
        "Zs": A[:, 1].copy(), "Ts": A[:, 5].copy(), "Ps": A[:, 4].copy(), "PLs": A[:, 3].copy(),
        "MFs_VAL": A[:, 6:8].copy() * 1e6,  # H2O, CO2 [ppmv]
        "MFs_ID": np.array([1, 2]),
    }


def synth_emissivities(seed=SEED_C4, n_emis=2000):
    """C4 emissivity set on the ASTER-DB knot axis: returns (X_e, emis[nK, nE])."""
    rng = np.random.default_rng(seed)
    X_e = np.linspace(1e4 / 14.5, 1e4 / 6.75, 791)
    P = rng.uniform(40.0, 400.0, n_emis)
    phi = rng.uniform(0.0, 2 * np.pi, n_emis)
    e = 0.85 + 0.1 * np.sin(2 * np.pi * X_e[:, None] / P[None, :] + phi[None, :]) \
        + 0.02 * rng.standard_normal((X_e.size, n_emis))
    return X_e, np.clip(e, 1e-4, 1 - 1e-4)


def synth_scene(seed=SEED_C5, n_pix=256 * 256, n_end=6, n_mix=2, Ts=287.87, dT=3.0, n_emis_db=2000):
    """C5 scene (after LWIR_HSI_Generator.py:147-162): n_end endmembers drawn from the C4 emissivity
    set, every pixel a mixture of n_mix of them with normalised uniform fractions and a surface temperature
    Ts + dT*N(0,1). Returns dict(end_idx [n_end] into the C4 set, kidx [n_pix][n_mix] in [0,n_end),
    frac [n_pix][n_mix], T [n_pix])."""
    rng = np.random.default_rng(seed)
    end_idx = rng.integers(0, n_emis_db, n_end)
    kidx = rng.integers(0, n_end, (n_pix, n_mix))
    frac = rng.random((n_pix, n_mix))
    frac /= frac.sum(axis=1)[:, None]
    T = Ts + dT * rng.standard_normal(n_pix)
    return {"end_idx": end_idx, "kidx": kidx.astype(np.int32), "frac": frac, "T": T}
