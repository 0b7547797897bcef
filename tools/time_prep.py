import os, sys, numpy as np, torch, ctypes as C
sys.path.insert(0, os.getcwd())
from radtxfr_amd import _lib, engine, synthetic
lib=_lib.load()
full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
a = synthetic.c3_atmosphere(32)
lines = engine.LineTable(full); grid = engine.Grid(500.0, 6000.0, 5500000)
w, p_atm = engine.layer_weights_od(lines.species, a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
q, m = engine.species_factors(lines.species, a["Ts"], weight=w)
plan = lines.plan(32, grid.n)
keep=[np.ascontiguousarray(x,dtype=np.float64) for x in (a["Ts"],p_atm,q,w,m)]
ev=[torch.cuda.Event(enable_timing=True) for _ in range(2)]
ts=[]
for it in range(8):
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(4):
        _lib.check(lib.rtx_line_prep(plan._h, lines._h, grid.byref(), 32, *[k.ctypes.data_as(C.c_void_p) for k in keep], 1.0,0.0,0.0,50.0,0.0,1.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    ev[1].record(); torch.cuda.synchronize(); ts.append(ev[0].elapsed_time(ev[1])/4)
print(os.path.basename(_lib.LIB_PATH), "prologue %.4f ms (min %.4f)"%(np.median(ts),min(ts)))
