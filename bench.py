#!/usr/bin/env python3
"""Headline benchmark: wavenumber x layer points / s of the line-by-line LWIR hot path on config C3
(BASELINE.json): 32-layer TUD (tau, L-up, L-down) from the standard atmosphere, 500-6000 cm^-1 at
0.001 cm^-1 (5.5 M wavenumbers), synthetic 100 000-line H2O+CO2 HITRAN-format table.

One step = one full pass for one atmosphere: per-atmosphere host factors (TIPS partition-sum ratios, column
weights) -> fp64 line prologue -> Voigt line-sum (OD[32][nX]) -> Planck + TUD integration (+ one RCCL all-gather of
tau/L-up/L-down when the wavenumber axis is sharded over GPUs). The line table is resident in HBM before the timed
region; everything that depends on the atmosphere is inside it.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement), with these extra objects:
  roofline     -- dominant kernel (the Voigt line-sum): algorithmic bytes / measured launch time vs 8 TB/s, PLUS the
                  roofline that actually binds it (`valu`: vector-instruction issue floor from the committed PMC
                  counts), and the other kernels' times, including the TUD kernel on an optically thin column
  cpu_baseline -- the NumPy oracle (port of the reference's CPU path) on a bounded sample: one core, and a process pool
                  over wavenumber chunks (the reference's multiprocessing.Pool pattern, Generate_LWIR_TUD.py:138-143)
  checksum     -- sums of the final tau / L-up / L-down (N > 1: of the all-gathered block). Shards are cut on line-sum
                  tile boundaries (dist.tile_aligned_bounds), so the spectra -- and these sums -- are IDENTICAL for every N
  sharding     -- (N > 1) per-rank shard lengths and line counts, per-rank kernel time of one step (min / max / all ranks:
                  the balance across ranks), `collective_ms` = one un-overlapped all-gather, `bytes_gathered`
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
N_SIMD = 256 * 4       # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # max shader clock (MI355X_MICROARCH.md); DVFS holds less under load, so the floor is optimistic
CYC_VALU, CYC_TRANS = 2.0, 8.0  # issue cycles per wave64 VALU instruction / per transcendental (same guide)
N_WAVENUMBERS = 5500000
N_LINES = 100000
N_LAYERS = 32
POOL_MAX = 16          # pool size of the CPU baseline. A one-GPU box lists 256 usable CPUs but its share is 16: measured on
                       # it (gpurun_out/r3/bench_1.json), 64 workers give 6.4e6 points/s in 1417 CPU-seconds -- each worker
                       # four times slower than the single-core leg -- against 6.8e6 with 16 workers


def _cpu_chunk(job):
    """One work unit of the CPU baseline: all 32 layers of the line-sum + the TUD integration on one wavenumber chunk.
    Runs in a pool worker (spawned: imports NumPy and the oracle only)."""
    i0, n = job
    from oracle import cpu_ref
    from radtxfr_amd import synthetic
    full = synthetic.synth_line_table(synthetic.SEED_C3, N_LINES, 475.0, 6025.0)
    atm = synthetic.c3_atmosphere(N_LAYERS)
    X = np.linspace(500.0, 6000.0, N_WAVENUMBERS)[i0:i0 + n]
    sub = synthetic.subset_table(full, X[0] - 12.0, X[-1] + 12.0)
    t0 = time.perf_counter()
    OD = np.stack([cpu_ref.layer_od(sub, X, atm["Ts"][k], atm["Ps"][k], atm["PLs"][k], atm["MFs_VAL"][k], atm["MFs_ID"])
                   for k in range(N_LAYERS)], axis=1)
    tau, Lu, Ld = cpu_ref.tud_from_od(X, OD, atm["Ts"], atm["Zs"])
    return time.perf_counter() - t0, int(sub["nu"].size), float(tau.sum())


def cpu_baseline():
    """Oracle (kind 'port') on bounded samples of the same workload. Leg 1: one process, a 160 cm^-1 window of the C3
    grid (all 32 layers, the lines that can reach it). Leg 2: the reference's own parallel pattern -- a process pool
    over independent work units -- with `cores` workers, each taking 40 cm^-1 chunks spread over the grid. Called
    before this process touches the GPU (workers are spawned children)."""
    import multiprocessing as mp
    n1, i1 = 160000, 2500000
    dt1, nl1, _ = _cpu_chunk((i1, n1))
    single = n1 * N_LAYERS / dt1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, POOL_MAX))
    nchunk, n2 = 2 * cores, 40000
    starts = np.linspace(0, N_WAVENUMBERS - n2, nchunk).astype(np.int64)
    with mp.get_context("spawn").Pool(cores) as pool:
        t0 = time.perf_counter()
        pool.map(_cpu_warm, range(cores), chunksize=1)  # interpreter start-up + imports of every worker: outside the timed wall
        t_start = time.perf_counter() - t0
        t0 = time.perf_counter()
        res = pool.map(_cpu_chunk, [(int(s), n2) for s in starts], chunksize=1)
        wall = time.perf_counter() - t0
    pooled = nchunk * n2 * N_LAYERS / wall
    return {"value": pooled, "unit": "wavenumber*layer points/s", "cores": cores, "kind": "port",
            "sample": f"{nchunk} chunks of {n2} wavenumbers spread over 500-6000 cm^-1 x {N_LAYERS} layers "
                      f"({nchunk * n2} of {N_WAVENUMBERS}), NumPy fp64 oracle, multiprocessing pool of {cores} "
                      f"(Generate_LWIR_TUD.py:138-143 pattern), {wall:.1f} s wall (worker start-up {t_start:.1f} s excluded), "
                      f"{sum(r[0] for r in res):.1f} s CPU",
            "single_core": {"value": single, "cores": 1,
                            "sample": f"{n1} of {N_WAVENUMBERS} wavenumbers (3000.0-3160.0 cm^-1) x {N_LAYERS} layers, "
                                      f"{nl1} lines in reach, {dt1:.1f} s"},
            "host_cpus": os.cpu_count(), "usable_cpus": avail}


def _cpu_warm(_):
    from oracle import cpu_ref  # noqa: F401
    from radtxfr_amd import synthetic  # noqa: F401
    return 0


def _latest_profile(suffix):
    d = os.path.join(ROOT, "profiles")
    cand = sorted(f for f in os.listdir(d) if f.endswith(suffix)) if os.path.isdir(d) else []
    if not cand:
        return None, None
    with open(os.path.join(d, cand[-1])) as fh:
        return json.load(fh), cand[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipelines", type=int, default=3,
                    help="independent pipelines (stream + records + OD buffer each) the atmospheres of the timed loop alternate over "
                         "(C3 on one GPU: 1.97 / 1.90 / 1.90 ms per step with 1 / 2 / 3; a rank's eighth of the grid: 0.30 / 0.27 / 0.25)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on a box with fewer GPUs than ranks (ranks share devices, gather staged through the host); never a measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()  # before the GPU is initialised: its workers are child processes

    import torch
    import torch.distributed as dist

    from radtxfr_amd import _lib, engine, synthetic

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    local = local % torch.cuda.device_count() if args.backend == "gloo" else local
    torch.cuda.set_device(local)
    lib = _lib.load()
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")

    # ---- inputs, resident in HBM before timing --------------------------------------------------
    full = synthetic.synth_line_table(synthetic.SEED_C3, N_LINES, 475.0, 6025.0)
    atm = synthetic.c3_atmosphere(N_LAYERS)
    grid_full = engine.Grid(500.0, 6000.0, N_WAVENUMBERS)
    # tile-aligned shards weighted by the per-tile cost estimate (dist.tud_shard_plan): every N computes the same bits,
    # and no rank is left with the expensive high-wavenumber end. `per` = the longest shard = the all-gather block.
    from radtxfr_amd import dist as rdist
    offs, reach = rdist.tud_shard_plan(full, 500.0, 6000.0, N_WAVENUMBERS, atm["Ts"], atm["Ps"], world)
    off, n_loc = int(offs[rank]), int(offs[rank + 1] - offs[rank])
    per = int(np.diff(offs).max())
    grid = grid_full.shard(off, n_loc)
    # each rank only needs the lines whose wings can reach its shard
    if world > 1:
        xs = (grid.x_at(0), grid.x_at(max(n_loc, 1) - 1))
        table = synthetic.subset_table(full, xs[0] - reach, xs[1] + reach)
    else:
        table = full
    lines = engine.LineTable(table)
    T, Z = atm["Ts"], atm["Zs"]
    dev = torch.device("cuda", local)
    OD = torch.empty((N_LAYERS, n_loc), dtype=torch.float32, device=dev)
    # N > 1: the TUD kernel writes straight into the packed [3][per] block that is all-gathered; two blocks so the
    # RCCL all-gather of step k overlaps the kernels of step k+1 (separate stream, async_op). The timed region ends
    # with every gather complete (drain + barrier + synchronize): throughput of a pipelined stream of atmospheres, the
    # reference's own use (199 atmospheres per run, Generate_LWIR_TUD.py:117-150), not the latency of one.
    NP = max(1, args.pipelines)
    gathered = [torch.empty((world * 3 * per,), dtype=torch.float32, device=dev) for _ in range(NP)] if world > 1 else None
    packed = [torch.zeros((3, per), dtype=torch.float32, device=dev) for _ in range(NP)] if world > 1 else None
    pending = [None] * NP
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    last = {}

    def host_factors(mf):
        # per-atmosphere host work of the path: column weights n(p,T) x_m PL and the TIPS ratios Q(296)/Q(T_k)
        w, p_atm = engine.layer_weights_od(lines.species, T, atm["Ps"], atm["PLs"], mf, atm["MFs_ID"])
        qratio, mass = engine.species_factors(lines.species, T, weight=w)
        return w, p_atm, qratio, mass

    # one call into the library per atmosphere (rtx_compute_tud): host factors + prologue + line-sum + TUD. The atmospheres
    # of the loop alternate over NP independent pipelines (engine.TudPipelines: stream + records + OD buffer each), so that
    # the prologue and the TUD pass of one atmosphere overlap the line-sum of the next: a stream of atmospheres is the
    # reference's own workload (199 per run, Generate_LWIR_TUD.py:117-150). N > 1: pipeline p writes straight into packed
    # block p, whose all-gather (async, RCCL's own stream) overlaps the kernels of the following atmospheres.
    pipes = engine.TudPipelines(lines, grid, Z, n_layers=N_LAYERS, n_pipes=NP,
                                outs=[(pk[0:1], pk[1:2], pk[2]) for pk in packed] if world > 1 else None) if n_loc > 0 else None
    runner = pipes.runs[0] if pipes is not None else None
    counter = [0]

    def step():
        if world == 1:
            last["p"], last["out"] = pipes.run(T, atm["Ps"], atm["PLs"], atm["MFs_VAL"], atm["MFs_ID"])
            return
        b = counter[0] % NP
        counter[0] += 1
        pk = packed[b]
        st_b = pipes.streams[b] if pipes is not None else torch.cuda.current_stream()
        with torch.cuda.stream(st_b):
            if pending[b] is not None:
                pending[b].wait()  # (stream b waits) this block's previous all-gather must be done before it is overwritten
                pending[b] = None
            if pipes is not None:
                pipes.k = b
                pipes.run(T, atm["Ps"], atm["PLs"], atm["MFs_VAL"], atm["MFs_ID"])
            if args.backend == "nccl":
                pending[b] = dist.all_gather_into_tensor(gathered[b], pk.view(-1), async_op=True)
            else:
                g_cpu = torch.empty(gathered[b].shape, dtype=gathered[b].dtype)
                dist.all_gather_into_tensor(g_cpu, pk.view(-1).cpu())
                gathered[b].copy_(g_cpu)
        last["b"] = b

    def drain():
        for b in range(NP):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def sync():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the interpreter's cyclic garbage collector is parked for the timed loop: a full collection takes 40-70 ms with torch's
    # object graph loaded (profiles/r3_time_dropin.txt) -- a hundred steps of an eighth-of-the-grid shard -- and says nothing about
    # the path; reference counting still frees everything the loop allocates
    import gc
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- checksum of the final spectra (N > 1: the all-gathered block, as every rank holds it) ----------------
    # formed the same way for every N -- one contiguous [3][nX] block -- so equal spectra give equal sums; `bits` adds the
    # float32 bit patterns as integers (exact, order-independent): equal iff the spectra are bit-identical
    if world > 1:
        g3 = gathered[last["b"]].view(world, 3, per)
        g = torch.cat([g3[r, :, :int(offs[r + 1] - offs[r])] for r in range(world)], dim=1).contiguous()
    else:
        g = torch.stack([last["out"][0][0], last["out"][1][0], last["out"][2]]).contiguous()
    assert g.shape == (3, N_WAVENUMBERS) and g.dtype == torch.float32
    sums = [float(g[c].double().sum()) for c in range(3)]
    bits = [int(g[c].view(torch.int32).to(torch.int64).sum().item()) for c in range(3)]

    # ---- N > 1: the balance across ranks and the collective, each measured on its own (outside `value`) ---------
    sharding = None
    if world > 1:
        NB = 4
        t_k = []
        for _ in range(3):  # NB steps back to back, no gather in between: this rank's kernels alone
            if runner is None:
                t_k.append(0.0)
                continue
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(NB):
                runner.run(T, atm["Ps"], atm["PLs"], atm["MFs_VAL"], atm["MFs_ID"])
            ev[1].record()
            torch.cuda.synchronize()
            t_k.append(ev[0].elapsed_time(ev[1]) / NB)
        cdev = dev if args.backend == "nccl" else "cpu"
        mine = torch.tensor([float(np.median(t_k)), float(n_loc), float(lines.n)], dtype=torch.float64, device=cdev)
        allr = torch.empty((world, 3), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr.view(-1), mine)
        allr = allr.cpu().numpy()
        t_c = []
        for _ in range(5):  # one all-gather with nothing else in flight, bracketed by barriers
            dist.barrier()
            torch.cuda.synchronize()
            if args.backend == "nccl":
                ev[0].record()
                dist.all_gather_into_tensor(gathered[0], packed[0].view(-1))
                ev[1].record()
                torch.cuda.synchronize()
                t_c.append(ev[0].elapsed_time(ev[1]))
            else:
                tc0 = time.perf_counter()
                g_cpu = torch.empty(gathered[0].shape, dtype=gathered[0].dtype)
                dist.all_gather_into_tensor(g_cpu, packed[0].view(-1).cpu())
                gathered[0].copy_(g_cpu)
                torch.cuda.synchronize()
                t_c.append((time.perf_counter() - tc0) * 1e3)
        tcoll = torch.tensor([float(np.median(t_c))], dtype=torch.float64, device=cdev)
        dist.all_reduce(tcoll, op=dist.ReduceOp.MAX)
        km = allr[:, 0]
        sharding = {"offsets": [int(v) for v in offs], "points_per_rank": [int(v) for v in allr[:, 1]],
                    "lines_per_rank": [int(v) for v in allr[:, 2]],
                    "kernel_ms": {"min": float(km.min()), "max": float(km.max()), "per_rank": [float(v) for v in km],
                                  "covers": "prologue + tile ranges + line-sum + TUD of this rank's shard, %d steps back to back, HIP events" % NB},
                    "collective_ms": float(tcoll.item()),
                    "collective": ("one RCCL all_gather_into_tensor, un-overlapped, HIP events on the launch stream, max over ranks"
                                   if args.backend == "nccl" else "gloo rehearsal: staged through the host, NOT a measurement"),
                    "bytes_gathered": int(world * 3 * per * 4), "bytes_per_rank": int(3 * per * 4),
                    "balance": "tile-aligned shards weighted by engine.tile_costs"}

    # ---- per-kernel launch time of the dominant kernel, HIP events on the launch stream ----------
    # (the same kernels through the three separate entry points, so that events can sit between the stages; NB launches
    # back to back per sample -- a synchronise per launch would put the launch latency inside the brackets)
    NB = 4
    t_voigt, t_tud = [], []
    import ctypes as C
    plan = lines.plan(N_LAYERS, n_loc) if n_loc > 0 else None
    for _ in range(min(max(args.steps, 3), 5)):
        if n_loc == 0:
            break
        w, p_atm, qratio, mass = host_factors(atm["MFs_VAL"])
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(NB):
            engine.voigt_sum(lines, grid, T, p_atm, w, out_f32=OD, qratio=qratio, mass=mass)
        ev[1].record()
        for _ in range(NB):
            engine.tud(OD, grid, T, Z)
        ev[2].record()
        torch.cuda.synchronize()
        t_voigt.append(ev[0].elapsed_time(ev[1]) / NB)  # prologue + tile ranges + line-sum kernel (+ its empty fp64 pass)
        t_tud.append(ev[1].elapsed_time(ev[2]) / NB)
    drain()
    # isolate the line-sum kernel: time the prologue alone, the same way, and subtract
    t_prep = []
    if n_loc > 0:
        w, p_atm, qratio, mass = host_factors(atm["MFs_VAL"])
        keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (T, p_atm, qratio, w, mass)]
        for _ in range(3):
            torch.cuda.synchronize()
            ev[0].record()
            for _ in range(NB):
                _lib.check(lib.rtx_line_prep(plan._h, lines._h, grid.byref(), N_LAYERS, *[k.ctypes.data_as(C.c_void_p) for k in keep],
                                             1.0, 0.0, 0.0, 50.0, 0.0, 1.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            ev[1].record()
            torch.cuda.synchronize()
            t_prep.append(ev[0].elapsed_time(ev[1]) / NB)
    ms_voigt = float(np.median(t_voigt) - np.median(t_prep)) if t_voigt else float("nan")
    ms_tud = float(np.median(t_tud)) if t_tud else float("nan")
    # the TUD kernel on an optically THIN column of the same grid (mixing ratios x 1e-3: tau spans (0,1) inside most
    # waves, every layer runs all 29 streams). The C3 column of SURVEY 8d is opaque almost everywhere, which lets the
    # kernel skip most streams; real LWIR atmospheres have windows. Outside `value`.
    ms_tud_thin = None
    if n_loc > 0:
        wt, p_t, q_t, m_t = host_factors(atm["MFs_VAL"] * 1e-3)
        engine.voigt_sum(lines, grid, T, p_t, wt, out_f32=OD, qratio=q_t, mass=m_t)
        out_t = (torch.empty((1, n_loc), dtype=torch.float32, device=dev), torch.empty((1, n_loc), dtype=torch.float32, device=dev),
                 torch.empty((n_loc,), dtype=torch.float32, device=dev))
        engine.tud(OD, grid, T, Z, out=out_t)
        tt_ = []
        for _ in range(3):  # four launches back to back per sample: the kernel, not the host's launch path
            ev[0].record()
            for _ in range(4):
                engine.tud(OD, grid, T, Z, out=out_t)
            ev[1].record()
            torch.cuda.synchronize()
            tt_.append(ev[0].elapsed_time(ev[1]) / 4.0)
        ms_tud_thin = float(np.median(tt_))

    if rank == 0:
        pts = float(N_WAVENUMBERS) * N_LAYERS
        # algorithmic bytes of one line-sum launch (SURVEY 8d stage A): 4 B OD write per point
        # + one 48 B fp32 line record per (line, layer)
        alg_bytes = 4.0 * n_loc * N_LAYERS + 48.0 * lines.n * N_LAYERS
        achieved = alg_bytes / (ms_voigt * 1e-3) / 1e9
        # counters cannot be collected inside this process (rocprofv3 passes): the committed measurement of this exact
        # workload (profiles/rN_pmc_*.json, written by tools/profile_round.sh + profile_summarize.py) is quoted
        traffic = tsrc = valu = None
        if world == 1:
            tj, tsrc = _latest_profile("_pmc_hbm_traffic.json")
            traffic = tj.get("hbm_bytes_per_launch") if tj else None
            vj, vsrc = _latest_profile("_pmc_valu.json")
            if vj:
                n_valu, n_trans = float(vj["sq_insts_valu"]), float(vj.get("sq_insts_valu_trans_f32") or 0.0)
                floor_ms = ((n_valu - n_trans) * CYC_VALU + n_trans * CYC_TRANS) / (N_SIMD * CLOCK_HZ) * 1e3
                valu = {"wave_instr_per_launch": n_valu, "transcendental_per_launch": n_trans or None,
                        "salu_per_launch": vj.get("sq_insts_salu"), "lds_per_launch": vj.get("sq_insts_lds"),
                        "issue_floor_ms": floor_ms, "frac": floor_ms / ms_voigt,
                        "model": f"{CYC_VALU:g} cycles per wave64 VALU instruction, {CYC_TRANS:g} per transcendental, "
                                 f"{N_SIMD} SIMDs at {CLOCK_HZ / 1e9:g} GHz",
                        "source": vsrc, "counted_on_ms_per_launch": vj.get("ms_per_launch")}
        name = ""
        try:
            buf = C.create_string_buffer(128)
            ncu = C.c_int(0)
            lib.rtx_device_info(buf, 128, C.byref(ncu))
            name = buf.value.decode()
        except Exception:
            pass
        out = {
            "metric": "wavenumber*layer points/s (32-layer TUD, 500-6000 cm^-1 @ 0.001 cm^-1)",
            "value": pts * args.steps / dt, "unit": "wavenumber*layer points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C3: 32-layer TUD (tau,Lu,Ld) StandardAtmosphere rows 1-32, 500-6000 cm^-1 @ 0.001 cm^-1",
                       "n_wavenumbers": N_WAVENUMBERS, "n_layers": N_LAYERS, "n_lines": N_LINES, "n_angles": 30,
                       "line_table": "synthetic HITRAN-format H2O+CO2, seed 20261005",
                       "parallelism": f"wavenumber-sharded x{world}" + (" + 1 RCCL all-gather" if world > 1 else ""),
                       "pipelines_per_gpu": NP},
            "roofline": {"kernel": "voigt_nodal_kernel",
                         # the kernel moves 1.1x its algorithmic bytes and is nowhere near HBM speed: what binds it is
                         # vector-instruction issue (SURVEY 8d stage A). `bound` names the roofline that achieved / peak /
                         # frac are quoted against (the contract's HBM one); `binding_unit` + `valu` say what it is actually up against.
                         "bound": "hbm", "binding_unit": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": tsrc if traffic is not None else None,
                         "ms_per_launch": ms_voigt, "algorithmic_bytes_per_launch": alg_bytes,
                         "ms_per_launch_covers": "HIP events around 4 back-to-back rtx_voigt_sum calls / 4: tile_ranges + tile_trim + "
                                                 "voigt_nodal_kernel<false> + its empty <true> instantiation (rocprofv3 times "
                                                 "the nodal kernel alone: profiles/*_bench_kernel_stats.csv)",
                         "valu": valu,
                         "note": "achieved/peak/frac are the HBM roofline the contract asks for (algorithmic bytes / "
                                 "launch time / 8 TB/s); the kernel is VALU-issue-bound, see `valu` and DESIGN.md 4.2",
                         "other_kernels_ms": {"line_prep_kernel": float(np.median(t_prep)), "tud_kernel": ms_tud,
                                              "tud_kernel_thin": ms_tud_thin,
                                              "tud_kernel_thin_note": "same grid, mixing ratios x1e-3 (tau spans (0,1)); outside `value`"}},
            "checksum": {"tau_sum": sums[0], "Lu_sum": sums[1], "Ld_sum": sums[2], "tau_bits": bits[0], "Lu_bits": bits[1],
                         "Ld_bits": bits[2], "note": "identical for every --gpus N (tile-aligned shards): *_bits is the exact integer sum "
                                                     "of the float32 bit patterns"},
            "device": name,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if sharding is not None:
            out["sharding"] = sharding
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    lines.close()


if __name__ == "__main__":
    main()
