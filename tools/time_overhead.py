"""Fixed cost of one C3-style step (host Python + ctypes + launches + small copies): the same calls as bench.py on a
shard so small that the kernels are negligible. This is the floor an 8-GPU strong-scaling step cannot go under.
    python tools/time_overhead.py [--n 20000] [--steps 300]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from radtxfr_amd import _lib, engine, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--steps", type=int, default=300)
args = ap.parse_args()
_lib.load()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
atm = synthetic.c3_atmosphere(32)
grid_full = engine.Grid(500.0, 6000.0, 5500000)
grid = grid_full.shard(2750000, args.n)
xs = grid.axis()
reach = engine.max_wing_cm(full, atm["Ts"], atm["Ps"] / 101325.0) + 1.0
table = synthetic.subset_table(full, xs[0] - reach, xs[-1] + reach)
lines = engine.LineTable(table)
T, Z = atm["Ts"], atm["Zs"]
w, p_atm = engine.layer_weights_od(lines.species, T, atm["Ps"], atm["PLs"], atm["MFs_VAL"], atm["MFs_ID"])
qratio, mass = engine.species_factors(lines.species, T)
OD = torch.empty((32, args.n), dtype=torch.float32, device="cuda")
pk = torch.zeros((3, args.n), dtype=torch.float32, device="cuda")

runner = engine.TudRunner(lines, grid, Z, n_layers=32, OD=OD, out=(pk[0:1], pk[1:2], pk[2]))


def step_separate():  # round-1 form: three entry points, factors precomputed outside
    engine.voigt_sum(lines, grid, T, p_atm, w, out_f32=OD, qratio=qratio, mass=mass)
    engine.tud(OD, grid, T, Z, out=(pk[0:1], pk[1:2], pk[2]))


def step():  # what bench.py runs per step: per-atmosphere host factors + ONE library call (rtx_compute_tud)
    runner.run(T, atm["Ps"], atm["PLs"], atm["MFs_VAL"], atm["MFs_ID"])


for fn, name in ((step_separate, "three calls, factors precomputed"),):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
    t_h = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name}: host-side {t_h / args.steps * 1e6:.0f} us per step")

for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"n={args.n} lines={lines.n}: host-side {t_host / args.steps * 1e6:.0f} us per step, with GPU drain {t_all / args.steps * 1e6:.0f} us per step")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
