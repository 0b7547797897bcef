"""`-m gpu`: the N > 1 path with the HIP engine actually running under every rank. The box has ONE GPU, so the ranks
are fresh child processes that share device 0 and talk over gloo (the driver's 8-GPU run uses RCCL; what is rehearsed
here is everything else: sharding, per-rank line subsets, grid offsets, packed blocks, reassembly, bench.py's
world > 1 branch). Children are started before they touch the GPU; nothing re-execs a GPU-initialised process."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_err
from oracle import cpu_ref as ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "mp_worker.py")
TOL_L, TOL_TAU = 1e-5, 2e-6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(mode, world, out_dir):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, WORKER, mode, str(r), str(world), str(port), str(out_dir)], env=env)
             for r in range(world)]
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=420))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert rcs == [0] * world, rcs


def test_two_ranks_sharded_compute_tud(tmp_path):
    """dist.compute_TUD_sharded on a C3 window, 3001 points over 2 ranks (tile-aligned shards, ragged last one, each
    rank holding only the lines in reach of its shard): every rank ends up with the full spectra; they are BIT-IDENTICAL
    to the single-rank HIP result (same tiles, candidate ranges trimmed to the reaching lines, same order of sums) and
    within 1e-5 / 2e-6 of the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mp_worker
    from radtxfr_amd import dist as rdist
    _run_ranks("tud", 2, tmp_path)
    lo, hi, dv, sub, a = mp_worker.tud_case()
    X1, tau1, Lu1, Ld1 = rdist.compute_TUD_sharded(lo, hi, dv, sub, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    Xr, tau_r, Lu_r, Ld_r = ref.compute_TUD(sub, lo, hi, dv, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert X1.size == 3001 and np.array_equal(X1, Xr) and tau_r.max() - tau_r.min() > 0.3
    for r in range(2):
        d = np.load(tmp_path / f"tud_r{r}.npz")
        assert np.array_equal(d["X"], Xr)
        for key, one in (("tau", tau1), ("Lu", Lu1), ("Ld", Ld1)):
            assert np.array_equal(d[key], one.cpu().numpy()), (key, float(np.max(np.abs(d[key] - one.cpu().numpy()))))
        assert np.max(np.abs(d["tau"] - tau_r)) <= TOL_TAU
        assert rel_err(d["Lu"], Lu_r) <= TOL_L and rel_err(d["Ld"], Ld_r) <= TOL_L
    d0, d1 = np.load(tmp_path / "tud_r0.npz"), np.load(tmp_path / "tud_r1.npz")
    assert all(np.array_equal(d0[k], d1[k]) for k in ("tau", "Lu", "Ld"))  # one all-gather: identical on every rank


def test_two_ranks_band_sharded_cube_end_to_end(tmp_path):
    """Config C5 end to end under 2 ranks (dist.hsi_cube_from_atmosphere: line table + atmosphere -> TUD on each rank's
    band span -> pixel cube -> one all-gather): bit-identical to the single-rank cube (tile-aligned shards, same
    lines, same order of sums) and within 1e-5 of the oracle, which forms every pixel's monochromatic spectrum and
    runs the reference's ILS on it (LWIR_HSI_Generator.py:151-167 + radiative_transfer.py:1072-1263)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mp_worker
    _run_ranks("cube", 2, tmp_path)
    xo1, cube1 = mp_worker.run_cube(False)  # this process: no process group -> world 1
    lo, hi, dv, sub, a, Xe, E, sc = mp_worker.cube_case()
    X, tau, Lu, Ld = ref.compute_TUD(sub, lo, hi, dv, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    assert 0.05 < np.median(tau) < 0.99
    E_hi = np.stack([np.interp(X, Xe, E[:, k]) for k in range(E.shape[1])], axis=1)
    em_p = np.einsum("pm,xpm->xp", sc["frac"], E_hi[:, sc["kidx"]])
    L = tau[:, None] * (em_p * ref.planckian(X, sc["T"]) + (1 - em_p) * Ld[:, None]) + Lu[:, None]
    xr, Lr = ref.ILS_MAKO(X, L, resFactor=2)
    assert np.array_equal(xo1, xr) and cube1.shape == Lr.shape and xr.size > 60
    assert rel_err(cube1, Lr) <= TOL_L
    for r in range(2):
        d = np.load(tmp_path / f"cube_r{r}.npz")
        assert np.array_equal(d["xo"], xr)
        assert np.array_equal(d["cube"], cube1), float(np.max(np.abs(d["cube"] - cube1)))


def test_bench_two_ranks_gloo_rehearsal(tmp_path):
    """bench.py's world > 1 branch (packed [3][per] blocks written by rtx_tud, double-buffered gather, barriers,
    max-over-ranks timing) executed end to end: `torch.distributed.run --nproc-per-node 2 bench.py --gpus 2
    --backend gloo`. A rehearsal on one shared GPU, never a measurement; the JSON line must be well-formed."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["n_wavenumbers"] == 5500000 and "checksum" in out and np.isfinite(out["checksum"]["tau_sum"])
    sh = out["sharding"]
    tile = 1024
    assert len(sh["offsets"]) == 3 and sh["offsets"][0] == 0 and sh["offsets"][2] == 5500000 and sh["offsets"][1] % tile == 0
    assert sum(sh["points_per_rank"]) == 5500000 and all(n < 100000 for n in sh["lines_per_rank"])  # per-rank line subsets
    assert sh["collective_ms"] > 0 and sh["bytes_gathered"] == 2 * 3 * max(sh["points_per_rank"]) * 4
    assert 0 < sh["kernel_ms"]["min"] <= sh["kernel_ms"]["max"] and len(sh["kernel_ms"]["per_rank"]) == 2
    # the same workload on ONE rank: identical spectra, bit for bit (integer sums of the float32 bit patterns)
    cmd1 = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    env1 = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    res1 = subprocess.run(cmd1, env=env1, capture_output=True, text=True, timeout=600)
    assert res1.returncode == 0, res1.stderr[-2000:]
    out1 = json.loads([ln for ln in res1.stdout.splitlines() if ln.startswith("{")][-1])
    for key in ("tau_bits", "Lu_bits", "Ld_bits", "tau_sum", "Lu_sum", "Ld_sum"):
        assert out["checksum"][key] == out1["checksum"][key], (key, out["checksum"][key], out1["checksum"][key])
    rec = os.environ.get("RADTXFR_REHEARSAL_RECORD")
    if rec:
        with open(rec, "w") as f:
            f.write(line + "\n")


def test_single_process_sharded_tud_and_local_comm():
    """ONE host process, several ranks (dist.LocalShardedTud + comm.LocalComm: rtx_comm_init_all / rtx_allgather): the
    reference's scripts have no launcher, so the wavenumber-sharded path must be reachable without torchrun. The box has one
    GPU: ranks share device 0 through the peer-copy backend (2 and 3 ranks, ragged and empty shards) and must reproduce the
    single-device spectra BIT FOR BIT for several atmospheres; RCCL itself is brought up on the one device (ncclCommInitAll
    through the run-time loader) and moves a block."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import mp_worker
    from radtxfr_amd import comm, dist as rdist
    lo, hi, dv, sub, a = mp_worker.tud_case()
    X1, tau1, Lu1, Ld1 = rdist.compute_TUD_sharded(lo, hi, dv, sub, a["Zs"], a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"], a["MFs_ID"])
    for devs in ([0, 0], [0, 0, 0], [0, 0, 0, 0, 0]):
        sh = rdist.LocalShardedTud(devs, lo, hi, dv, sub, a["Zs"], a["Ts"], a["Ps"])
        assert sh.comm.backend == "peer" and sh.offs[-1] == 3001 and all(v % 1024 == 0 for v in sh.offs[1:-1] if v < 3001)
        for scale in (1.0, 0.5):
            X, tau, Lu, Ld = sh.run(a["Ts"], a["Ps"], a["PLs"], a["MFs_VAL"] * scale, a["MFs_ID"])
            torch.cuda.synchronize()
            if scale == 1.0:
                assert np.array_equal(X, X1)
                for got, one in ((tau, tau1), (Lu, Lu1), (Ld, Ld1)):
                    assert torch.equal(got, one), (devs, float((got - one).abs().max()))
            else:
                assert float(tau.mean()) > float(tau1.mean())
        sh.close()
    c = comm.LocalComm([0])
    assert c.backend == "rccl", "librccl was not found by the run-time loader"
    send = torch.arange(1000, dtype=torch.float32, device="cuda")
    recv = torch.zeros(1000, dtype=torch.float32, device="cuda")
    c.all_gather([send], [recv])
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    c.close()
    with pytest.raises(Exception):
        comm.LocalComm([0, 0], backend=1)  # RCCL refuses a repeated device
