"""Import harness for the upstream reference (TEST INFRASTRUCTURE ONLY).

Only used in the build container, where /root/reference exists, by the scripts
that generate committed data (oracle/make_tables.py, oracle/make_golden.py).
Nothing under radtxfr_amd/, bench.py, or the `-m gpu` tests imports this.

Bypasses documented in SURVEY.md section 8c (ordinary NumPy-2 breakages of the
reference, not environment denials):
  1. rt.make_spectral_axis passes a float `num` to np.linspace
     (radiative_transfer.py:269-270)  -> replaced by an int-cast twin.
  2. rt.compute_OD needs the absent LBLRTM binary (radiative_transfer.py:395-501)
     -> replaced by `oracle_OD`, which drives the reference's own
     hapi.absorptionCoefficient_Voigt (misc/hapi.py:10906) per molecule.
  3. hapi.arange_ float-`num` bug (misc/hapi.py:133-139) -> always pass OmegaGrid=.
"""
import contextlib
import io
import os
import sys

REF = "/root/reference"


def load():
    """Return (rt, hapi, ils_gauss_module) imported from the read-only reference."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference checkout not present (expected only in the build container)")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    for p in (os.path.join(REF, "misc"), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    with contextlib.redirect_stdout(io.StringIO()):
        import hapi  # noqa: E402  (prints a banner)
        import radiative_transfer as rt  # noqa: E402
        import ILS_MAKO as ils_gauss  # noqa: E402
    import numpy as np

    rt.make_spectral_axis = lambda a, b, d: np.linspace(a, b, int(np.ceil((b - a) / d)))
    return rt, hapi, ils_gauss


def inject_table(hapi, name, tbl):
    """Put a synthetic HITRAN-format table into hapi.LOCAL_TABLE_CACHE (misc/hapi.py:438-463)."""
    n = len(tbl["nu"])
    hapi.LOCAL_TABLE_CACHE[name] = {
        "header": {"number_of_rows": n, "table_name": name},
        "data": {k: [v.item() for v in tbl[k]] for k in tbl},
    }


def make_oracle_OD(hapi, table_name, tbl):
    """compute_OD stand-in with the SURVEY 8(a-3) semantics, built on the reference's own line-sum."""
    import numpy as np

    pairs = sorted(set(zip(tbl["molec_id"].tolist(), tbl["local_iso_id"].tolist())))

    def oracle_OD(Xmin, Xmax, opts=None, T=296.0, P=101325.0, PL=1.0, MF_VAL=None, MF_ID=None, DVOUT=None, **kw):
        dv = DVOUT if DVOUT is not None else opts["DVOUT"]
        X = np.linspace(Xmin, Xmax, int(np.ceil((Xmax - Xmin) / dv)))
        od = np.zeros_like(X)
        for m, ppmv in zip(np.asarray(MF_ID).tolist(), np.asarray(MF_VAL).tolist()):
            comps = [(mm, ii) for (mm, ii) in pairs if mm == m]
            if not comps:
                continue
            with contextlib.redirect_stdout(io.StringIO()):
                _, xs = hapi.absorptionCoefficient_Voigt(
                    Components=comps, SourceTables=table_name,
                    Environment={"T": float(T), "p": float(P) / 101325.0},
                    OmegaGrid=X, HITRAN_units=False)
            od += xs * (ppmv * 1e-6) * PL * 1e5
        return X, od

    return oracle_OD
