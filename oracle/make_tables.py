"""Extract the published physical constant tables the line-sum needs, as DATA.

Run once in the build container (needs /root/reference):
    python oracle/make_tables.py
writes radtxfr_amd/data/tips2011.npz with

  tdat     (119,)    TIPS-2011 temperature nodes 60:25:3010 K   (misc/hapi.py:5401-5413)
  mi       (n,2)     (molecule, local isotopologue) ids that have a TIPS table
  q        (n,119)   total internal partition sums Q(T) at the nodes; the reference
                     holds them as float32 constants (misc/hapi.py:5418-9565)
  iso_mi   (k,2)     (M,I) ids of the ISO table                    (misc/hapi.py:3372-3496)
  iso_abun (k,)      natural abundances
  iso_mass (k,)      molar masses [g/mol]

These are numbers published with HITRAN/TIPS-2011 (Fischer & Gamache), re-expressed
as arrays; no reference code is copied. The .npz travels to the GPU box.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _refimport import load  # noqa: E402


def main():
    _, hapi, _ = load()
    keys = sorted(hapi.TIPS_ISO_HASH.keys())
    tdat = np.asarray(hapi.Tdat, dtype=np.float64)
    q = np.zeros((len(keys), tdat.size), dtype=np.float32)
    for r, k in enumerate(keys):
        v = np.asarray(hapi.TIPS_ISO_HASH[k])
        assert v.dtype == np.float32, (k, v.dtype)
        # a few isotopologues carry a 1-element placeholder (no TIPS-2011 data): NaN row
        q[r] = v if v.size == tdat.size else np.nan
    ik = sorted(hapi.ISO.keys())
    ia = np.array([hapi.ISO[k][hapi.ISO_INDEX["abundance"]] for k in ik], dtype=np.float64)
    im = np.array([hapi.ISO[k][hapi.ISO_INDEX["mass"]] for k in ik], dtype=np.float64)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "radtxfr_amd", "data", "tips2011.npz")
    np.savez_compressed(out, tdat=tdat, mi=np.array(keys, dtype=np.int32), q=q,
                        iso_mi=np.array(ik, dtype=np.int32), iso_abun=ia, iso_mass=im)
    print("wrote", os.path.normpath(out), "TIPS rows", len(keys), "ISO rows", len(ik))


if __name__ == "__main__":
    main()
