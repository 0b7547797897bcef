"""Child process of the multi-rank `-m gpu` tests (tests/test_gpu_multirank.py): one rank of a world-size-N gloo group
whose ranks all run the HIP engine on the one GPU of the box. Started fresh (python tests/mp_worker.py ...), so it
initialises the GPU itself; never re-exec'd from a GPU-initialised process.

    python tests/mp_worker.py <mode> <rank> <world> <port> <out_dir>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def tud_case():
    """A C3 window with a ragged last shard (3001 points over 2 ranks), thinned so tau spans (0,1)."""
    from radtxfr_amd import synthetic
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi, dv = 1000.0, 1003.001, 0.001  # ceil(3.001/0.001) = 3001 points
    sub = synthetic.subset_table(full, lo - 15.0, hi + 15.0)
    a = synthetic.c3_atmosphere(8)
    a["MFs_VAL"] = a["MFs_VAL"] * 1e-3
    return lo, hi, dv, sub, a


def cube_case(n_pix=96):
    """C5 at oracle size: 8-layer column on 900-1100 cm^-1 at 0.01 cm^-1, 96 mixed pixels, resFactor-2 MAKO bands."""
    from radtxfr_amd import synthetic
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    lo, hi, dv = 900.0, 1100.0, 0.01
    sub = synthetic.subset_table(full, lo - 15.0, hi + 15.0)
    a = synthetic.c3_atmosphere(8)
    a["MFs_VAL"] = a["MFs_VAL"] * 3e-3
    Xe, em = synthetic.synth_emissivities(n_emis=2000)
    sc = synthetic.synth_scene(n_pix=n_pix)
    return lo, hi, dv, sub, a, Xe, em[:, sc["end_idx"]], sc


def run_cube(group_ready):
    import torch
    from radtxfr_amd import dist as rdist
    lo, hi, dv, sub, a, Xe, E, sc = cube_case()
    dev = torch.device("cuda", 0)
    f32 = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32), device=dev)
    xo, cube = rdist.hsi_cube_from_atmosphere(lo, hi, dv, sub, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"], Xe,
                                              f32(E), torch.as_tensor(sc["kidx"], device=dev), f32(sc["frac"]),
                                              torch.as_tensor(sc["T"], device=dev), resFactor=2)
    torch.cuda.synchronize()
    return xo, cube.cpu().numpy()


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from radtxfr_amd import dist as rdist
        if mode == "tud":
            lo, hi, dv, sub, a = tud_case()
            X, tau, Lu, Ld = rdist.compute_TUD_sharded(lo, hi, dv, sub, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
            torch.cuda.synchronize()
            np.savez(os.path.join(out, f"tud_r{rank}.npz"), X=X, tau=tau.cpu().numpy(), Lu=Lu.cpu().numpy(), Ld=Ld.cpu().numpy())
        elif mode == "cube":
            xo, cube = run_cube(True)
            np.savez(os.path.join(out, f"cube_r{rank}.npz"), xo=xo, cube=cube)
        else:
            raise SystemExit("unknown mode " + mode)
        with open("/proc/self/maps") as f:
            assert any("libradtxfr_hip.so" in ln for ln in f), "HIP library not mapped in the child"
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
