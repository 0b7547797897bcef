"""Device float32 spectra -> host NumPy arrays for the drop-in functions (radiative_transfer.compute_TUD and
compute_TUD_batch return float64 NumPy arrays, as the reference does, radiative_transfer.py:388-392).

At C3 size one result (tau, L-up, L-down as float64) is 132 MB; the device computes it in ~2 ms, so how it reaches the
host decides what a caller sees. Two ways, both through page-locked memory (pageable hipMemcpy is several times slower):

  zero-copy   widened to float64 on the DEVICE, one asynchronous copy into a pinned block, the arrays handed out are
              views of that block; when they die the block returns to this module's idle list and the next result
              reuses it. Nothing touches the data on the host: 2.7 ms per result. A result that is KEPT keeps its block
              page-locked, so the blocks owned this way are capped (PINNED_RESULT_CAP); when all are held, and for
              batches that keep every result, the second way is used.
  pageable    float32 crosses PCIe into a small ring of reusable pinned staging buffers (half the bytes), and a pool of
              host threads widens it into ordinary pageable arrays (NumPy releases the GIL in its copy loops; a fresh
              132 MB array costs ~35 ms of page faults and copying on one thread).
"""
import os
import threading
import weakref
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

PINNED_RESULT_CAP = int(os.environ.get("RADTXFR_PINNED_RESULT_BYTES", str(1 << 30)))  # pinned bytes owned for zero-copy results
_lent = [0]          # bytes currently handed out
_owned = [0]         # bytes of every block this module has page-locked (handed out + idle)
_idle = {}           # (rows, n) -> [pinned float64 tensors] waiting for reuse
_lent_lock = threading.RLock()  # re-entrant: a garbage collection under the lock may run a result's finaliser (_release)
_pool = None
_CHUNK = 1 << 19  # elements per widening task (2-4 MB)


def _threads():
    global _pool
    if _pool is None:
        try:
            n = len(os.sched_getaffinity(0))
        except AttributeError:
            n = os.cpu_count() or 1
        _pool = ThreadPoolExecutor(max_workers=max(1, min(8, n)), thread_name_prefix="radtxfr-hostio")
    return _pool


def pinned_lent_bytes():
    """Pinned bytes currently handed out as zero-copy results (for tests and diagnostics)."""
    return _lent[0]


def _release(host, nbytes):
    """A result's arrays have all died: its block goes back to the idle list (never unpinned: page-locking 132 MB costs
    ~10 ms, and PyTorch's own pinned allocator showed 40-90 ms stalls when blocks of this size came and went)."""
    with _lent_lock:
        _lent[0] -= nbytes
        _idle.setdefault(tuple(host.shape), []).append(host)


def _pinned_block(shape):
    """An idle block of this shape, or a new one while the module owns less than PINNED_RESULT_CAP; else None."""
    nbytes = shape[0] * shape[1] * 8
    with _lent_lock:
        if _lent[0] + nbytes > PINNED_RESULT_CAP:
            return None
        free = _idle.get(shape)
        if free:
            _lent[0] += nbytes
            return free.pop()
        # make room for a block of a new shape by dropping idle ones of other shapes
        while _owned[0] + nbytes > PINNED_RESULT_CAP:
            victim = next((k for k, v in _idle.items() if v), None)
            if victim is None:
                return None
            t = _idle[victim].pop()
            _owned[0] -= t.numel() * 8
        _owned[0] += nbytes
        _lent[0] += nbytes
    try:
        return torch.empty(shape, dtype=torch.float64, pin_memory=True)
    except Exception:  # page-locking failed: the bytes were never owned
        with _lent_lock:
            _owned[0] -= nbytes
            _lent[0] -= nbytes
        raise


def lend_block(shape):
    """(pinned float64 tensor [rows][n], its NumPy view) lent out as a zero-copy result, or None when the cap is reached.
    The block returns to the idle list when the NumPy array and every view of it have died."""
    host = _pinned_block((int(shape[0]), int(shape[1])))
    if host is None:
        return None
    arr = host.numpy()
    weakref.finalize(arr, _release, host, int(shape[0]) * int(shape[1]) * 8)
    return host, arr


def rows_to_pinned_f64(rows, stream=None):
    """float32 device rows [(k_i, n)] -> float64 NumPy views of ONE pinned block (zero-copy path), or None when every
    block the cap allows is still held by earlier results. Returns (arrays, event): valid once `event` has completed."""
    n = rows[0].shape[-1]
    ks = [int(np.prod(r.shape[:-1])) if r.dim() > 1 else 1 for r in rows]
    nbytes = sum(ks) * n * 8
    host = _pinned_block((sum(ks), n))
    if host is None:
        return None
    dev_block = torch.empty((sum(ks), n), dtype=torch.float64, device=rows[0].device)
    o = 0
    for r, k in zip(rows, ks):
        dev_block[o:o + k].copy_(r.reshape(k, n))
        o += k
    side = stream if stream is not None else torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record()  # the widening runs on the current (compute) stream
    with torch.cuda.stream(side):
        side.wait_event(ready)
        host.copy_(dev_block, non_blocking=True)
        dev_block.record_stream(side)
        done = torch.cuda.Event()
        done.record(side)
    arr = host.numpy()
    weakref.finalize(arr, _release, host, nbytes)  # every view handed out keeps `arr` alive; then the block is idle again
    out, o = [], 0
    for k in ks:
        out.append(arr[o:o + k])
        o += k
    return out, done


class Staging:
    """A ring of reusable pinned float32 staging buffers of one pipeline (pageable path). stage() enqueues the
    device-to-host copy of a result into the next slot; collect() waits for it and widens it into fresh pageable arrays
    on the host thread pool. The caller collects a ticket before `depth` further results have been staged (the slot is
    overwritten then)."""

    def __init__(self, depth=2):
        self.depth = depth
        self.bufs = [None] * depth
        self.k = 0

    def stage(self, rows, stream=None):
        n = rows[0].shape[-1]
        ks = [int(np.prod(r.shape[:-1])) if r.dim() > 1 else 1 for r in rows]
        slot = self.k % self.depth
        self.k += 1
        buf = self.bufs[slot]
        if buf is None or buf.shape[0] < sum(ks) or buf.shape[1] != n or buf.dtype != rows[0].dtype:
            buf = self.bufs[slot] = torch.empty((sum(ks), n), dtype=rows[0].dtype, pin_memory=True)
        side = stream if stream is not None else torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            o = 0
            for r, k in zip(rows, ks):
                buf[o:o + k].copy_(r.reshape(k, n), non_blocking=True)
                r.record_stream(side)
                o += k
            done = torch.cuda.Event()
            done.record(side)
        return (slot, ks, n, done)

    def collect(self, ticket, dtype=np.float64):
        """Wait for the copy, widen into pageable arrays of `dtype` with the thread pool; returns the list of arrays
        (one [k_i][n] array per staged row block) once every chunk is written."""
        slot, ks, n, done = ticket
        done.synchronize()
        src = self.bufs[slot].numpy()
        outs = [np.empty((k, n), dtype=dtype) for k in ks]
        futs = []
        pool = _threads()
        o = 0
        for out, k in zip(outs, ks):
            for r in range(k):
                for c0 in range(0, n, _CHUNK):
                    futs.append(pool.submit(np.copyto, out[r, c0:c0 + _CHUNK], src[o + r, c0:c0 + _CHUNK], "same_kind"))
            o += k
        for f in futs:
            f.result()
        return outs


def copy_threaded(a):
    """A writable copy of a large contiguous 1-D array made by the thread pool (fresh pages are faulted in parallel)."""
    out = np.empty_like(a)
    pool = _threads()
    futs = [pool.submit(np.copyto, out[c0:c0 + _CHUNK], a[c0:c0 + _CHUNK]) for c0 in range(0, a.size, _CHUNK)]
    for f in futs:
        f.result()
    return out
