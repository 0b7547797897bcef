#!/usr/bin/env python3
"""Upper bound of what fusing the line-sum into the TUD kernel (OD never written to HBM; SURVEY 8d row "A+B fused")
could save on C3, measured instead of estimated: the same two kernels on wavenumber chunks small enough that the layer
optical depths of a chunk (n x 32 x 4 B) stay in the 256 MB Infinity Cache between the line-sum that writes them and the
TUD kernel that reads them. If the per-point times of the cache-resident chunks equal those of the full grid, neither
kernel is waiting on the OD traffic and a fused kernel has nothing to gain from removing it.
    python tools/time_fused_bound.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, synthetic

_lib.load()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
atm = synthetic.c3_atmosphere(32)
lines = engine.LineTable(full)
grid_full = engine.Grid(500.0, 6000.0, 5500000)
T, Z = atm["Ts"], atm["Zs"]
tile = int(_lib.load().rtx_voigt_tile_points())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
for mf in (1.0, 1e-3):
    w, p_atm = engine.layer_weights_od(lines.species, T, atm["Ps"], atm["PLs"], atm["MFs_VAL"] * mf, atm["MFs_ID"])
    q, m = engine.species_factors(lines.species, T, weight=w)
    for nchunk in (1, 4, 16, 64):
        n = (5500000 // nchunk) // tile * tile
        OD = torch.empty((32, n), dtype=torch.float32, device="cuda")
        tv = tt = 0.0
        reps = 3
        for rep in range(reps + 1):
            for c in range(nchunk):
                g = grid_full.shard(c * n, n)
                ev[0].record()
                engine.voigt_sum(lines, g, T, p_atm, w, out_f32=OD, qratio=q, mass=m)
                ev[1].record()
                engine.tud(OD, g, T, Z)
                ev[2].record()
                torch.cuda.synchronize()
                if rep:
                    tv += ev[0].elapsed_time(ev[1])
                    tt += ev[1].elapsed_time(ev[2])
        pts = nchunk * n * reps
        print(f"mf x{mf:g}: {nchunk:3d} chunks of {n:8d} points (OD of a chunk {n * 128 / 1e6:7.1f} MB): "
              f"prologue+line-sum {tv / pts * 5.5e6:.3f} ms, TUD {tt / pts * 5.5e6:.3f} ms per 5.5 M points")
        del OD
lines.close()
