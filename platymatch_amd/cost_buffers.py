"""The cost buffer a large registration writes its matrices into, kept per (device, stream) between calls and handed out as a LEASE
(never two registrations in one buffer); the early lease that asks for it on a helper thread at the start of a registration.
pipeline.py decides WHEN a buffer is kept (COST_CACHE_MIN_BYTES) and how large it must be; this module only owns the storage."""
import threading

import numpy as np

# The eight cost matrices of a large registration (160 GB at 50 000 x 50 000) are kept by THIS module between calls, one buffer
# per (device, stream), instead of being handed back to torch's caching allocator: a freed block of that size is the only one
# large enough for any later request of more than a megabyte that finds no exact fit, gets split for it, and — with a live
# piece inside — can neither serve the next registration nor be returned to the driver (round 3: the fifth 50 000-point
# registration of a process ran out of memory with 149 GB "reserved but unallocated").  Buffers below COST_CACHE_MIN_BYTES
# (and every registration of a batch) go through the allocator as before.  release_cost_buffers() gives the memory back.
_COST_CACHE = {}
_COST_LOCK = threading.RLock()        # (re-entrant: an allocation that fails inside cost_buffer() releases the other streams' idle buffers)


def _cost_key(device):
    import torch
    return (device.index, torch.cuda.current_stream(device).cuda_stream)


class _CostLease:
    """A registration's hold on the kept cost buffer of its (device, stream): release() — always from a finally — makes the
    buffer available to the next registration.  A buffer is handed to ONE registration at a time: two host threads registering
    on the same stream (e.g. both on the default stream) must not write their matrices into the same storage while the other's
    assignment passes still read it."""

    def __init__(self, key, view):
        self.key, self.view = key, view

    def release(self):
        with _COST_LOCK:
            e = _COST_CACHE.get(self.key)
            if e is not None and e["lease"] is self:
                e["lease"] = None
        self.view = None


def cost_buffer(device, shape):
    """Lease a float64 [shape] view of this (device, stream)'s kept buffer, grown if it is too small (the old one is released
    first).  -> _CostLease, or None when another registration holds the buffer: the caller then takes a fresh allocation."""
    import torch
    from .device_memory import big_empty
    need = int(np.prod(shape))
    key = _cost_key(device)
    with _COST_LOCK:
        e = _COST_CACHE.get(key)
        if e is not None and e["lease"] is not None:
            return None
        if e is None or e["t"].numel() < need:
            _COST_CACHE.pop(key, None)
            e = None                                       # (released before the larger one is asked for)
            # (a raw block, device_memory.py: released = back with the driver at once, never split by torch for something small)
            e = {"t": big_empty((need,), torch.float64, device, keep=False), "lease": None}
            _COST_CACHE[key] = e
        lease = _CostLease(key, e["t"][:need].view(*shape))
        e["lease"] = lease
    return lease


def kept_cost_bytes(device):
    """Bytes of this (device, stream)'s kept buffer that a new registration can have (0 while another one holds it)."""
    with _COST_LOCK:
        e = _COST_CACHE.get(_cost_key(device))
        return 0 if (e is None or e["lease"] is not None) else e["t"].numel() * 8


def release_cost_buffers():
    """Hand every kept cost buffer that no registration holds back to torch's allocator (and, with torch.cuda.empty_cache(), to
    the driver); a buffer in use goes when its registration releases it... the next call of this function."""
    with _COST_LOCK:
        for key in [k for k, e in _COST_CACHE.items() if e["lease"] is None]:
            del _COST_CACHE[key]


class _EarlyLease:
    """The kept cost buffer asked for on a helper thread AT THE START of a registration: a fresh process pays
    ~22 ms per GB for the buffer's first allocation (0.9 s for the default mode's 40 GB at 50 000 nuclei, 3.5 s for the exact
    mode's 160 GB), and until the cost kernel needs it the host is busy with other first-call costs — self-check, code-object
    loads, statistics, descriptors.  result() -> the lease, or None (another registration holds the buffer / allocation failed:
    the caller goes the ordinary way).  A buffer that already exists is leased on the spot, without a thread."""

    def __init__(self, device, nbytes):
        import torch
        self.lease, self.thread, self.nbytes = None, None, int(nbytes)
        stream = torch.cuda.current_stream(device)
        elems = (int(nbytes) + 7) // 8
        if kept_cost_bytes(device) >= 8 * elems:
            self.lease = cost_buffer(device, (elems,))
            return

        def work():
            try:
                with torch.cuda.device(device), torch.cuda.stream(stream):      # (the buffer is kept per (device, stream))
                    self.lease = cost_buffer(device, (elems,))
            except Exception:       # noqa: BLE001 — out of memory here is not an error: the caller decides again with what is free
                self.lease = None

        self.thread = threading.Thread(target=work, name="pm-cost-buffer")
        self.thread.start()

    def result(self):
        if self.thread is not None:
            self.thread.join()
            self.thread = None
        return self.lease

    def cancel(self):
        lease = self.result()
        if lease is not None:
            lease.release()
        self.lease = None
