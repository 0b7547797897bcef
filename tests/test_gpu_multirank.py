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
    """dist.compute_TUD_sharded on a C3 window, 3001 points over 2 ranks (ragged last shard): every rank ends up with
    the full spectra; they equal the single-rank HIP result (1e-6: a different tiling regroups fp32 sums) and the
    oracle (1e-5 / 2e-6)."""
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
        assert np.max(np.abs(d["tau"] - tau1.cpu().numpy())) <= 1e-6
        assert rel_err(d["Lu"], Lu1.cpu().numpy()) <= 1e-6 and rel_err(d["Ld"], Ld1.cpu().numpy()) <= 1e-6
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
    rec = os.environ.get("RADTXFR_REHEARSAL_RECORD")
    if rec:
        with open(rec, "w") as f:
            f.write(line + "\n")
