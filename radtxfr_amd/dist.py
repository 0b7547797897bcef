"""Wavenumber sharding across the GPUs of one node: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests), ONE collective per spectrum.

Every stage of the hot path is pointwise in wavenumber except the line windows, so the monochromatic
grid splits into contiguous chunks with no halo of *data*: each rank keeps only the lines whose wings
can reach its chunk (engine.max_wing_cm) and evaluates grid values bit-identical to the full grid
(rtx_grid carries the global offset). The only exchange is the final all-gather of the packed
[tau, L-up, L-down] chunk (3 x nX/G floats per rank: 8.25 MB at G = 8 on the C3 grid).

Shards are cut on LINE-SUM TILE boundaries of the full axis (tile_aligned_bounds): a rank's tiles are then
tiles of the single-rank run, its candidate ranges are trimmed to the lines that reach each tile
(csrc/rtx_voigt.hip), and its results are bit-identical to the single-rank run for any number of ranks.
Shard lengths are weighted by a per-tile cost estimate (engine.tile_costs: lines in reach, line centres and
Weideman band rows per tile, all layers), because equal lengths leave the rank with the highest wavenumbers
-- widest Doppler cores, most band rows -- as the straggler of every step.

The reference's own parallel axis is multiprocessing.Pool over atmospheres (Generate_LWIR_TUD.py:117-150):
independent replicas, radiative_transfer.compute_TUD_batch(devices=...) from one process.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_total, world, rank):
    """Contiguous equal chunks of ceil(n_total/world) items (the last rank may be short or empty): (offset, count,
    per). Used for axes with no tile structure (MAKO bands, rows)."""
    per = (int(n_total) + world - 1) // world
    off = min(rank * per, int(n_total))
    return off, max(0, min(per, int(n_total) - off)), per


def tile_aligned_bounds(n_total, world, tile, cost=None):
    """Offsets [world + 1] of contiguous wavenumber shards: offs[0] = 0, offs[world] = n_total, every interior
    offset a multiple of `tile` (the line-sum tile, rtx_voigt_tile_points()). cost [n_tiles] (any positive
    per-tile weights, e.g. engine.tile_costs) balances the shards' summed cost; None = equal tile counts.
    Deterministic: every rank computes the same cut from the same inputs. Trailing shards may be empty when
    there are fewer tiles than ranks."""
    n_total, world, tile = int(n_total), int(world), int(tile)
    n_tiles = (n_total + tile - 1) // tile
    c = np.ones(n_tiles) if cost is None else np.maximum(np.asarray(cost, dtype=np.float64).ravel(), 0.0)
    assert c.size == n_tiles, (c.size, n_tiles)
    if not np.any(c > 0):
        c = np.ones(n_tiles)
    cum = np.concatenate([[0.0], np.cumsum(c)])
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        j = int(np.searchsorted(cum, target, side="left"))  # first j with cum[j] >= target
        if j > 0 and target - cum[j - 1] < cum[min(j, n_tiles)] - target:
            j -= 1  # the nearer of the two neighbouring tile boundaries
        cuts.append(min(max(j, cuts[-1]), n_tiles))
    offs = np.minimum(np.asarray(cuts + [n_tiles], dtype=np.int64) * tile, n_total)
    offs[-1] = n_total
    return offs


def subset_lines(columns, x_lo, x_hi, reach):
    """Rows of a line table (column dict) whose centre lies within `reach` cm^-1 of [x_lo, x_hi]."""
    nu = np.asarray(columns["nu"])
    m = (nu >= x_lo - reach) & (nu <= x_hi + reach)
    return {k: np.asarray(v)[m] for k, v in columns.items()}


def _all_gather_flat(send, world, group=None):
    """ONE all_gather_into_tensor of equal flat blocks. RCCL ("nccl") gathers device buffers in place over xGMI; the
    gloo backend (CPU tests, and rehearsals of the N > 1 path with several ranks sharing one GPU) has no device
    all-gather, so device blocks are staged through the host there."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        recv = torch.empty((world * send.numel(),), dtype=send.dtype)
        dist.all_gather_into_tensor(recv, send.cpu(), group=group)
        return recv.to(send.device)
    recv = torch.empty((world * send.numel(),), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv


def all_gather_spectra(local, n_total, group=None, offs=None):
    """local: [C][n_loc] tensor (C stacked spectra of this rank's chunk, any device the backend supports).
    Returns [C][n_total] on every rank. One all_gather_into_tensor of a padded [C][per] block. offs [world + 1]:
    the shard offsets (tile_aligned_bounds); None = shard_bounds' equal chunks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local[:, :n_total]
    rank = dist.get_rank(group)
    if offs is None:
        per0 = (int(n_total) + world - 1) // world
        offs = np.minimum(np.arange(world + 1, dtype=np.int64) * per0, int(n_total))
    offs = np.asarray(offs, dtype=np.int64)
    lens = np.diff(offs)
    per = int(lens.max())
    n_loc = int(lens[rank])
    assert local.shape[1] == n_loc, (local.shape, n_loc)
    C = local.shape[0]
    send = local if n_loc == per else torch.nn.functional.pad(local, (0, per - n_loc))
    send = send.contiguous().view(-1)  # flat buffers: accepted by both the RCCL and the gloo backends
    from .engine import trace_range
    with trace_range("all_gather_spectra"):
        recv = _all_gather_flat(send, world, group).view(world, C, per)
    if np.all(lens[:-1] == per):  # equal chunks, ragged tail only: one strided view
        return recv.permute(1, 0, 2).reshape(C, world * per)[:, :n_total]
    return torch.cat([recv[r, :, :int(lens[r])] for r in range(world)], dim=1)


def sharded_tud(compute_local, columns, Xmin, Xmax, n_total, reach, group=None, offs=None):
    """Run `compute_local(sub_table, offset, n_loc) -> [3][n_loc] tensor (tau, Lu, Ld)` on this rank's
    chunk of np.linspace(Xmin, Xmax, n_total) and reassemble the full spectra on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if offs is None:
        per0 = (int(n_total) + world - 1) // world
        offs = np.minimum(np.arange(world + 1, dtype=np.int64) * per0, int(n_total))
    off, n_loc = int(offs[rank]), int(offs[rank + 1] - offs[rank])
    step = (Xmax - Xmin) / (n_total - 1)
    x_lo, x_hi = Xmin + off * step, Xmin + (off + max(n_loc, 1) - 1) * step
    sub = subset_lines(columns, x_lo, x_hi, reach)
    local = compute_local(sub, off, n_loc)
    return all_gather_spectra(local, n_total, group, offs)


def tud_shard_plan(line_table, Xmin, Xmax, n_total, Ts, Ps, world, balance=True):
    """(offs [world + 1], reach [cm^-1]) for a wavenumber-sharded compute_TUD: tile-aligned shards of the axis
    np.linspace(Xmin, Xmax, n_total), weighted by engine.tile_costs when `balance`; `reach` = how far outside its
    shard a rank must keep line centres. Pure host arithmetic on the table's columns: identical on every rank."""
    from . import _lib, engine
    tile = int(_lib.load().rtx_voigt_tile_points())
    p_atm = np.asarray(Ps, dtype=np.float64) / 101325.0
    step = (Xmax - Xmin) / (n_total - 1)
    reach = engine.max_wing_cm(line_table, Ts, p_atm) + step
    cost = engine.tile_costs(line_table, Xmin, step, n_total, Ts, p_atm, tile) if (balance and world > 1) else None
    return tile_aligned_bounds(n_total, world, tile, cost), reach


def compute_TUD_sharded(Xmin, Xmax, DVOUT, line_table, Zs, Ts, Ps, PLs, MFs_VAL, MFs_ID, Altitudes=(500,), theta_r=0.0,
                        N_angle=30, group=None, balance=True):
    """compute_TUD (radiative_transfer.py:274-392) with the spectral axis sharded over the ranks of
    `group`; every rank returns the full (X, tau, Lu, Ld) as float32 device tensors (X as NumPy fp64).
    One sensor altitude / slant path per call. Shards are tile-aligned (tud_shard_plan): the spectra are
    bit-identical for every world size, including 1."""
    from . import engine
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_total = int(np.ceil((Xmax - Xmin) / DVOUT))
    grid_full = engine.Grid(Xmin, Xmax, n_total)
    offs, reach = tud_shard_plan(line_table, Xmin, Xmax, n_total, Ts, Ps, world, balance)

    def compute_local(sub, off, n_loc):
        dev = engine.device()
        if n_loc == 0:
            return torch.empty((3, 0), dtype=torch.float32, device=dev)
        grid = grid_full.shard(off, n_loc)
        lines = engine.LineTable(sub)
        try:
            OD = engine.optical_depths(lines, grid, Ts, Ps, PLs, MFs_VAL, MFs_ID)
            tau, Lu, Ld, _ = engine.tud(OD, grid, Ts, Zs, Altitudes=Altitudes, theta_r=theta_r, N_angle=N_angle)
            return torch.stack([tau[0], Lu[0], Ld])
        finally:
            torch.cuda.synchronize()
            lines.close()

    full = sharded_tud(compute_local, line_table, Xmin, Xmax, n_total, reach, group, offs)
    return np.linspace(Xmin, Xmax, n_total), full[0], full[1], full[2]


class LocalShardedTud:
    """compute_TUD with the spectral axis sharded over several GPUs by ONE host process (no torchrun, no process group):
    the same tile-aligned, cost-weighted shards as compute_TUD_sharded -- hence the same bits as a single-device run -- one
    line-table subset, runner and stream per device, and one comm.LocalComm.all_gather (RCCL over xGMI, or peer copies) of
    the packed [tau, L-up, L-down] blocks. Set up once per (grid, table, altitude grid); run() per atmosphere is
    asynchronous apart from the per-atmosphere host factors.

        sh = LocalShardedTud(devices, Xmin, Xmax, DVOUT, table, Zs, Ts, Ps)   # Ts, Ps: a typical atmosphere, for the plan
        X, tau, Lu, Ld = sh.run(Ts, Ps, PLs, MFs_VAL, MFs_ID)                  # float32 views on devices[0]

    The tensors run() returns are views of this object's buffers, overwritten by the next run(): copy what must be kept.
    One sensor altitude and slant path per object (the packed block carries three rows)."""

    def __init__(self, devices, Xmin, Xmax, DVOUT, line_table, Zs, Ts, Ps, Altitudes=(500,), theta_r=0.0, N_angle=30, balance=True,
                 backend=-1):
        from . import comm, engine
        self.devices = [int(d) for d in devices]
        world = len(self.devices)
        self.n_total = int(np.ceil((Xmax - Xmin) / DVOUT))
        self.Xmin, self.Xmax = float(Xmin), float(Xmax)
        grid_full = engine.Grid(Xmin, Xmax, self.n_total)
        with torch.cuda.device(self.devices[0]):
            self.offs, reach = tud_shard_plan(line_table, Xmin, Xmax, self.n_total, Ts, Ps, world, balance)
        self.per = int(np.diff(self.offs).max())
        self.comm = comm.LocalComm(self.devices, backend) if world > 1 else None
        self.ranks = []
        step = grid_full.step
        for r, d in enumerate(self.devices):
            off, n_loc = int(self.offs[r]), int(self.offs[r + 1] - self.offs[r])
            with torch.cuda.device(d):
                st = torch.cuda.Stream()
                packed = torch.zeros((3, self.per), dtype=torch.float32, device="cuda")
                gathered = torch.empty((world * 3 * self.per,), dtype=torch.float32, device="cuda") if world > 1 else None
                run = lines = None
                if n_loc > 0:
                    grid = grid_full.shard(off, n_loc)
                    sub = subset_lines(line_table, Xmin + off * step, Xmin + (off + n_loc - 1) * step, reach)
                    lines = engine.LineTable(sub)
                    with torch.cuda.stream(st):
                        run = engine.TudRunner(lines, grid, Zs, n_layers=np.asarray(Ts).size, Altitudes=Altitudes, theta_r=theta_r,
                                               N_angle=N_angle, out=(packed[0:1], packed[1:2], packed[2]),
                                               plan=engine.VoigtPlan(lines, np.asarray(Ts).size, n_loc))
            self.ranks.append(dict(dev=d, stream=st, packed=packed, gathered=gathered, run=run, lines=lines))

    def run(self, Ts, Ps, PLs, MFs_VAL, MFs_ID):
        world = len(self.devices)
        for rk in self.ranks:
            if rk["run"] is not None:
                with torch.cuda.device(rk["dev"]), torch.cuda.stream(rk["stream"]):
                    rk["run"].run(Ts, Ps, PLs, MFs_VAL, MFs_ID)
        r0 = self.ranks[0]
        if world == 1:
            full = r0["packed"][:, :self.n_total]
        else:
            self.comm.all_gather([rk["packed"].view(-1) for rk in self.ranks], [rk["gathered"] for rk in self.ranks],
                                 [rk["stream"] for rk in self.ranks])
            with torch.cuda.device(r0["dev"]), torch.cuda.stream(r0["stream"]):
                g3 = r0["gathered"].view(world, 3, self.per)
                full = torch.cat([g3[r, :, :int(self.offs[r + 1] - self.offs[r])] for r in range(world)], dim=1)
        torch.cuda.current_stream(r0["dev"]).wait_stream(r0["stream"])  # the caller's stream on devices[0] sees the result
        return np.linspace(self.Xmin, self.Xmax, self.n_total), full[0], full[1], full[2]

    def close(self):
        for rk in self.ranks:
            with torch.cuda.device(rk["dev"]):
                torch.cuda.synchronize()
                if rk["run"] is not None:
                    rk["run"].plan.close()
                if rk["lines"] is not None:
                    rk["lines"].close()
        self.ranks = []
        if self.comm is not None:
            self.comm.close()
            self.comm = None


def all_gather_rows(local, n_rows_total, group=None):
    """local: [n_loc][M] tensor holding this rank's shard_bounds() slice of the first axis.
    Returns [n_rows_total][M] on every rank (one all_gather_into_tensor of a padded block)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local[:n_rows_total]
    rank = dist.get_rank(group)
    _, n_loc, per = shard_bounds(n_rows_total, world, rank)
    assert local.shape[0] == n_loc, (local.shape, n_loc)
    M = local.shape[1]
    send = local if n_loc == per else torch.cat([local, local.new_zeros((per - n_loc, M))])
    recv = _all_gather_flat(send.contiguous().view(-1), world, group)
    return recv.view(world * per, M)[:n_rows_total]


def hsi_cube_sharded(compute_bands, n_bands_total, group=None):
    """Config C5 across GPUs: the cube is cut along the BAND axis (a band's triangle only needs the
    wavenumbers under it, so a band shard is a wavenumber shard with a ~3 % halo that is recomputed, not
    exchanged). `compute_bands(b0, b1) -> [b1-b0][nPix]` tensor, e.g.
        lambda b0, b1: sensor.hsi_cube(grid, tau, La, Ld, Xk, E, kidx, frac, T, band_slice=(b0, b1))[1]
    One all-gather assembles the [n_bands_total][nPix] cube on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b0, nb, _ = shard_bounds(n_bands_total, world, rank)
    return all_gather_rows(compute_bands(b0, b0 + nb), n_bands_total, group)


def hsi_cube_from_atmosphere(Xmin, Xmax, DVOUT, line_table, Zs, Ts, Ps, PLs, MFs_VAL, MFs_ID, Xk, endmembers, kidx, frac, Tpix,
                             resFactor=2, Altitudes=(500,), theta_r=0.0, N_angle=30, group=None):
    """Config C5 end to end across the ranks of `group` (the reference's single-process model is
    LWIR_HSI_Generator.py:109-179 on a stored TUD): line table + atmosphere -> monochromatic tau, L-up, L-down on the
    MAKO span -> per-pixel at-sensor radiance -> triangle ILS -> cube [n_bands][n_pixels].

    The cube is cut along the BAND axis. Rank r owns bands [b0, b1) and computes the whole path -- prologue, line-sum,
    TUD, band moments, pixel cube -- ONLY on the wavenumbers under those bands' triangles (centre +- sigma); neighbouring
    ranks recompute the ~3 % of points their edge triangles share instead of exchanging them. ONE all-gather assembles
    the cube on every rank. Shards start on a line-sum tile boundary of the full axis and every rank keeps the whole
    line table of the span, so each rank's numbers are bit-identical to the single-rank run (same tiles, same
    candidate lines, same summation order).

    Xk [nk] knot axis, endmembers [nk][nEnd] float32, kidx [nPix][nMix] int32, frac [nPix][nMix] float32, Tpix [nPix]
    float64: device tensors (sensor.hsi_cube). Returns (X_out [nB] NumPy, cube [nB][nPix] float32 device)."""
    from . import _lib, engine, sensor
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_total = int(np.ceil((Xmax - Xmin) / DVOUT))
    grid_full = engine.Grid(Xmin, Xmax, n_total)
    X_out, centre, sigma = sensor.mako_bands(grid_full.x_at(0), grid_full.x_at(n_total - 1), resFactor)
    nB = X_out.size
    b0, nb, _ = shard_bounds(nB, world, rank)
    dev = endmembers.device
    if nb == 0:
        local = torch.empty((0, kidx.shape[0]), dtype=torch.float32, device=dev)
    else:
        tile = int(_lib.load().rtx_voigt_tile_points())
        x_lo = float(np.min(centre[b0:b0 + nb] - sigma[b0:b0 + nb]))
        x_hi = float(np.max(centre[b0:b0 + nb] + sigma[b0:b0 + nb]))
        i_lo = max(0, int(np.floor((x_lo - Xmin) / grid_full.step)) - 1)
        i_lo -= i_lo % tile  # start on a tile boundary of the full axis ...
        i_hi = int(np.ceil((x_hi - Xmin) / grid_full.step)) + 2
        i_hi = min(n_total, i_hi + (-i_hi) % tile)  # ... and end on one (or at the end of the axis)
        grid = grid_full.shard(i_lo, i_hi - i_lo)
        lines = line_table if isinstance(line_table, engine.LineTable) else engine.LineTable(line_table)
        try:
            OD = engine.optical_depths(lines, grid, Ts, Ps, PLs, MFs_VAL, MFs_ID)
            tau, Lu, Ld, _ = engine.tud(OD, grid, Ts, Zs, Altitudes=Altitudes, theta_r=theta_r, N_angle=N_angle)
            local = sensor.hsi_cube(grid, tau[0], Lu[0], Ld, Xk, endmembers, kidx, frac, Tpix, resFactor=resFactor,
                                    bands=(X_out[b0:b0 + nb], centre[b0:b0 + nb], sigma[b0:b0 + nb]))[1]
        finally:
            if lines is not line_table:
                torch.cuda.synchronize()
                lines.close()
    return X_out, all_gather_rows(local, nB, group)
