#!/usr/bin/env python3
"""Per-workgroup and per-XCD time spread of the line-sum on the C3 grid, from the debug build that stamps every wave's
lifetime (tools/build_variant.sh stamp -DRTX_SC_STAMP=1):

    RADTXFR_LIB=build/stamp.so python tools/tile_spread.py [--table clustered]

Prints, for the nodal kernel's launch: the distribution of workgroup lifetimes (s_memtime ticks, 100 MHz), the busiest tiles
with their candidate counts, the summed lifetime per XCD (workgroup b runs on XCD b & 7), and what the slowest single
workgroup is as a share of the launch -- a hot tile that serialises a launch shows up there."""
import argparse, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--table", default="uniform", choices=["uniform", "clustered"])
ap.add_argument("--layers", type=int, default=32)
args = ap.parse_args()
stamp = os.path.join(tempfile.gettempdir(), "rtx_stamp.bin")
os.environ["RADTXFR_STAMP_FILE"] = stamp
import torch
from radtxfr_amd import _lib, engine, synthetic
lib = _lib.load()
full = (synthetic.synth_clustered_table if args.table == "clustered" else synthetic.synth_line_table)(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(args.layers)
lines = engine.LineTable(full)
N = 5500000
grid = engine.Grid(500.0, 6000.0, N)
tile = int(lib.rtx_voigt_tile_points())
OD = torch.empty((args.layers, N), dtype=torch.float32, device="cuda")
w, p_atm = engine.layer_weights_od(lines.species, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for _ in range(2):
    ev[0].record()
    engine.voigt_sum(lines, grid, a["Ts"], p_atm, w, out_f32=OD)
    ev[1].record()
    torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1])
if not os.path.exists(stamp):
    raise SystemExit("no stamp file: run with RADTXFR_LIB pointing at a -DRTX_SC_STAMP=1 build")
raw = np.fromfile(stamp, dtype=np.uint64)
NW, NS = 2, 8
n_tiles = (N + tile - 1) // tile
slots = raw.size // (args.layers * NW * NS)  # grid.x
st = raw.reshape(args.layers, slots, NW, NS)
life = st[..., 7].max(axis=2).astype(np.float64)  # workgroup lifetime = its slower wave  [layer][slot]
RTX_XCD_CHUNK = 16
b = np.arange(slots)
idx, xcd = b >> 3, b & 7
c, wq = idx // RTX_XCD_CHUNK, idx % RTX_XCD_CHUNK
tile_of = (c * 8 + xcd) * RTX_XCD_CHUNK + wq
valid = tile_of < n_tiles
lv = life[:, valid]
print(f"{args.table} table: prologue + line-sum (stamped build, with the stamp read-back) {ms:.2f} ms; {valid.sum()} tiles x {args.layers} layers")
q = np.percentile(lv, [50, 90, 99, 99.9, 100])
print("workgroup lifetime [ticks]: median %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f  (max / median = %.1f)" % (*q, q[4] / q[0]))
per_xcd = np.array([life[:, (xcd == x) & valid].sum() for x in range(8)])
print("summed lifetime per XCD / mean:", " ".join("%.3f" % v for v in per_xcd / per_xcd.mean()))
tot = lv.sum()
conc = 256 * 4 * 6 // NW  # resident workgroups: 256 CUs x 4 SIMDs x 6 waves / 2 waves per workgroup
print("ideal launch = total lifetime / %d resident workgroups = %.0f ticks; slowest workgroup = %.0f ticks = %.1f %% of that"
      % (conc, tot / conc, q[4], 100.0 * q[4] / (tot / conc)))
nu = full["nu"]
k_hot, s_hot = np.unravel_index(np.argsort(life, axis=None)[::-1][:5], life.shape)
for k, s in zip(k_hot, s_hot):
    t = tile_of[s]
    x0 = 500.0 + t * tile * grid.step
    cand = int(np.sum((nu > x0 - 6.0) & (nu < x0 + tile * grid.step + 6.0)))
    print(f"  layer {k:2d} tile {t:5d} ({x0:8.2f} cm^-1): {life[k, s]:.0f} ticks, ~{cand} lines within 6 cm^-1")
lines.close()
