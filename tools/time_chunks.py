import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from radtxfr_amd import engine, synthetic, radiative_transfer as rt
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
lines = engine.LineTable(full)
for nl, dv in ((32, 0.001), (66, 0.0005)):
    A = synthetic.load_standard_atmosphere()[:nl]
    a = dict(Zs=A[:, 1], Ts=A[:, 5], Ps=A[:, 4], PLs=A[:, 3], MFs_VAL=A[:, 6:8] * 1e6, MFs_ID=np.array([1, 2]))
    for k in (1, 2, 4, 6, 8, 12):
        ts = []
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = rt.compute_TUD(500.0, 6000.0, DVOUT=dv, line_table=lines, chunks=k, **a)
            ts.append((time.perf_counter() - t0) * 1e3); del r
        print(f"{nl} layers, DVOUT {dv}: chunks={k}: {np.median(ts[2:]):.2f} ms per call", flush=True)
