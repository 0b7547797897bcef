"""CPU, world_size 2, gloo: the N>1 path of radtxfr_amd.dist -- contiguous wavenumber shards, each rank
keeping only the lines that can reach its shard, ONE all-gather to reassemble (tau, Lu, Ld).
The per-rank compute is stood in for by the oracle (no GPU here); the sharding, the line subsetting
(engine.max_wing_cm) and the reassembly are the code under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref as ref
from radtxfr_amd import dist as rdist
from radtxfr_amd import engine, synthetic

LO, HI, N = 1000.0, 1003.0, 3001  # odd on purpose: ragged last shard


def _case():
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    sub = synthetic.subset_table(full, LO - 15.0, HI + 15.0)
    a = synthetic.c3_atmosphere(8)
    a["MFs_VAL"] = a["MFs_VAL"] * 1e-3  # optically thin enough that tau is not ~0 everywhere
    return sub, a


def _oracle(table, X, a):
    OD = np.stack([ref.layer_od(table, X, a["Ts"][k], a["Ps"][k], a["PLs"][k], a["MFs_VAL"][k], a["MFs_ID"])
                   for k in range(a["Ts"].size)], axis=1)
    return ref.tud_from_od(X, OD, a["Ts"], a["Zs"])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table, a = _case()
        X = np.linspace(LO, HI, N)
        reach = engine.max_wing_cm(table, a["Ts"], a["Ps"] / 101325.0) + (HI - LO) / (N - 1)
        seen = {}

        def compute_local(sub, off, n_loc):
            seen["n_lines"] = sub["nu"].size
            tau, Lu, Ld = _oracle(sub, X[off:off + n_loc], a)
            return torch.from_numpy(np.stack([tau, Lu, Ld]))

        full = rdist.sharded_tud(compute_local, table, LO, HI, N, reach)
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full.numpy())
        np.save(os.path.join(out_dir, f"n{rank}.npy"), np.array([seen["n_lines"], table["nu"].size]))
    finally:
        dist.destroy_process_group()


def _cube_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(7 * 5, dtype=torch.float32).reshape(7, 5)  # 7 "bands" over 2 ranks: ragged
        got = rdist.hsi_cube_sharded(lambda b0, b1: full[b0:b1].clone(), 7)
        np.save(os.path.join(out_dir, f"c{rank}.npy"), got.numpy())
    finally:
        dist.destroy_process_group()


def test_world2_gloo_band_sharded_cube(tmp_path):
    mp.spawn(_cube_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = np.arange(35, dtype=np.float32).reshape(7, 5)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"c{r}.npy"), want)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds():
    for n, w in ((5500000, 8), (3001, 2), (7, 4), (3, 8)):
        got = [rdist.shard_bounds(n, w, r) for r in range(w)]
        assert sum(g[1] for g in got) == n
        assert all(got[r][0] == sum(g[1] for g in got[:r]) or got[r][1] == 0 for r in range(w))


def test_tile_aligned_bounds():
    """Interior offsets on tile boundaries, monotone, covering [0, n); a cost vector moves the cuts towards equal cost."""
    for n, w, tile in ((5500000, 8, 1024), (3001, 2, 1024), (3001, 8, 1024), (700, 4, 1024), (1024 * 16, 4, 1024)):
        o = rdist.tile_aligned_bounds(n, w, tile)
        assert o.shape == (w + 1,) and o[0] == 0 and o[-1] == n and np.all(np.diff(o) >= 0)
        assert all(v % tile == 0 for v in o[1:-1] if v < n)
    n, tile = 5500000, 1024
    nt = (n + tile - 1) // tile
    cost = np.linspace(1.0, 3.0, nt)  # cost rising with wavenumber: shards must shrink towards the top
    o = rdist.tile_aligned_bounds(n, 8, tile, cost)
    lens = np.diff(o)
    assert np.all(np.diff(lens) < 0)
    share = np.array([cost[o[r] // tile:(o[r + 1] + tile - 1) // tile].sum() for r in range(8)]) / cost.sum()
    assert np.max(np.abs(share - 0.125)) < 1e-3
    assert np.array_equal(o, rdist.tile_aligned_bounds(n, 8, tile, cost))  # deterministic


def test_tile_costs_follow_the_table():
    """engine.tile_costs: more lines in reach -> more cost; Weideman band rows (high wavenumbers, low pressure) cost
    extra; the C3 plan is tile-aligned and its shards shrink with wavenumber."""
    full = synthetic.synth_line_table(synthetic.SEED_C3, 100000, 475.0, 6025.0)
    a = synthetic.c3_atmosphere(8)
    n = 5500000
    step = 5500.0 / (n - 1)
    c = engine.tile_costs(full, 500.0, step, n, a["Ts"], a["Ps"] / 101325.0, 1024)
    assert c.shape == ((n + 1023) // 1024,) and np.all(c > 0)
    assert c[-500:].mean() > c[:500].mean()  # wider Doppler cores at 6000 cm^-1
    dense = {k: np.concatenate([v, v[(full["nu"] > 1000) & (full["nu"] < 1010)]]) for k, v in full.items()}
    order = np.argsort(dense["nu"], kind="stable")
    dense = {k: v[order] for k, v in dense.items()}
    c2 = engine.tile_costs(dense, 500.0, step, n, a["Ts"], a["Ps"] / 101325.0, 1024)
    t0 = int((1005.0 - 500.0) / step / 1024)
    assert c2[t0] > 1.3 * c[t0] and abs(c2[10] - c[10]) < 1e-9 * c[10]
    offs, reach = rdist.tud_shard_plan(full, 500.0, 6000.0, n, a["Ts"], a["Ps"], 8)
    assert offs[0] == 0 and offs[-1] == n and all(v % 1024 == 0 for v in offs[1:-1]) and reach > 5.0
    assert np.all(np.diff(np.diff(offs)) < 0)


def _uneven_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        offs = np.array([0, 2048, 3001])
        full = torch.arange(3 * 3001, dtype=torch.float32).reshape(3, 3001)
        got = rdist.all_gather_spectra(full[:, offs[rank]:offs[rank + 1]].clone(), 3001, offs=offs)
        np.save(os.path.join(out_dir, f"u{rank}.npy"), got.numpy())
    finally:
        dist.destroy_process_group()


def test_world2_gloo_uneven_tile_aligned_shards(tmp_path):
    """all_gather_spectra with shards of different lengths (tile-aligned cuts): padded to the longest, one all-gather,
    reassembled by the offsets."""
    mp.spawn(_uneven_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = np.arange(3 * 3001, dtype=np.float32).reshape(3, 3001)
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"u{r}.npy"), want)


def test_world2_gloo_sharded_tud(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    table, a = _case()
    X = np.linspace(LO, HI, N)
    want = np.stack(_oracle(table, X, a))
    assert want[0].min() < 0.9 and want[0].max() > 0.1  # the case exercises real transmittances
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got.shape == (3, N)
        # shards see fewer lines than the full table, yet nothing that reaches them is dropped
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
        nl = np.load(tmp_path / f"n{r}.npy")
        assert nl[0] < nl[1]


def test_all_gather_single_process_identity():
    x = torch.arange(12.0).reshape(3, 4)
    assert torch.equal(rdist.all_gather_spectra(x, 4), x)
