"""Device memory for the path's MATRICES: blocks of BIG_BLOCK_BYTES and more come straight from the driver (pm_device_alloc,
include/platymatch_hip.h) and go back to it whole; everything smaller stays with torch's caching allocator.

Why: the eight N x M cost matrices of the reference (_dock_widget.py:547-602) are 160 GB at 50 000 nuclei, a filter matrix 10-160
GB, a transposed copy or a fallback pairing's two exact matrices 20-40 GB.  torch's allocator keeps a freed block of that size and
SPLITS it for the next request of a megabyte or more that finds no exact fit; with one long-lived piece inside (a solver workspace,
a descriptor array) the block can neither serve the next matrix nor go back to the driver.  Round 5's large-size soak — registrations
of 30 000 to 52 000 nuclei of VARYING size in one process — ran out of memory in a third of its cases with 140 GB "reserved but
unallocated" (tools/auto_soak.py --min-points).  Blocks taken here are never split: an idle one serves a request of exactly its
size on the stream it was used on (the common case: the same registration again), idle blocks beyond MAX_IDLE_FRACTION of the
device go back to the driver oldest first, all of them when an allocation fails — then the unleased kept cost buffers and torch's
own idle memory too — before the request is tried once more and, failing again, raises torch.OutOfMemoryError.

A raw allocation costs what torch's first allocation of that size costs (the driver maps and wipes the pages: 25-55 ms per GB on
this pool, tools/cold_start.py), which is why idle blocks are kept at all."""
import collections
import ctypes
import threading

import numpy as np

from . import _native as nat

BIG_BLOCK_BYTES = 4 << 30          # matrices from here on bypass torch's allocator (a batch's 2k-20k registrations stay below: they
                                   # come and go in many sizes side by side, which is what a splitting allocator is good at)
MAX_IDLE_FRACTION = 0.25           # of the device's memory: idle blocks kept for the next request of their size
_GRANULE = 2 << 20

_LOCK = threading.RLock()
_IDLE = collections.OrderedDict()  # ptr -> (device index, stream, bytes), oldest first
_TOTAL = {}                        # device index -> bytes of HBM
_STATS = collections.Counter()     # "raw_allocations", "reused", "returned", "trims": what the tests and tools read


_THREAD = threading.local()


class through_torch:
    """with through_torch(): this thread's big_empty calls go to torch's allocator whatever their size — for a BATCH of
    registrations (batch.py): many pairs of many sizes side by side is what a splitting, caching allocator is good at, and a raw
    block per pair would pay the driver's 25-55 ms per GB on every one of them."""

    def __enter__(self):
        _THREAD.depth = getattr(_THREAD, "depth", 0) + 1

    def __exit__(self, *exc):
        _THREAD.depth -= 1


class _Block:
    """One raw allocation, alive as long as any tensor views it (torch keeps the object behind __cuda_array_interface__)."""

    def __init__(self, ptr, nbytes, dev_index, stream, keep):
        self.ptr, self.nbytes, self.dev_index, self.stream, self.keep = ptr, nbytes, dev_index, stream, keep
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}

    def __del__(self):
        try:
            _retire(self.ptr, self.nbytes, self.dev_index, self.stream, self.keep)
        except Exception:          # noqa: BLE001 — interpreter shutdown: the driver reclaims the memory with the process
            pass


def _resolve(device):
    """-> torch.device with an explicit index ('cuda' alone means the current device)."""
    torch = nat.torch_mod()
    dev = nat.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _free(dev_index, ptr):
    nat.check(nat.load().pm_device_free(dev_index, ctypes.c_void_p(ptr)))
    _STATS["returned"] += 1


def _retire(ptr, nbytes, dev_index, stream, keep):
    """The last view of a block is gone (any thread, the host side of its stream has issued all its work): keep it idle or free it."""
    drop = []
    with _LOCK:
        if keep:
            _IDLE[ptr] = (dev_index, stream, nbytes)
            cap = MAX_IDLE_FRACTION * _TOTAL.get(dev_index, 0)
            while sum(b for d, _, b in list(_IDLE.values()) if d == dev_index) > cap:      # (list(): a collector-run __del__ may re-enter)
                old = next(p for p, (d, _, _) in list(_IDLE.items()) if d == dev_index)      # oldest of this device
                drop.append((old, _IDLE.pop(old)[0]))
        else:
            drop.append((ptr, dev_index))
    for p, d in drop:
        _free(d, p)


def idle_bytes(device=None):
    """Bytes of idle raw blocks (of one device): room a new allocation can draw on, like torch's reserved-but-unused memory."""
    idx = None if device is None else _resolve(device).index
    with _LOCK:
        return sum(b for d, _, b in list(_IDLE.values()) if idx is None or d == idx)


def trim(device=None):
    """Every idle raw block (of one device) back to the driver -> bytes given back."""
    idx = None if device is None else _resolve(device).index
    with _LOCK:
        gone = [(p, v) for p, v in list(_IDLE.items()) if idx is None or v[0] == idx]
        for p, _ in gone:
            _IDLE.pop(p, None)
        _STATS["trims"] += 1
    for p, (d, _, _) in gone:
        _free(d, p)
    return sum(v[2] for _, v in gone)


def release_everything_idle(device=None):
    """What an allocation failure calls for: idle raw blocks, the kept cost buffers no registration holds, torch's idle memory."""
    torch = nat.torch_mod()
    from .cost_buffers import release_cost_buffers
    release_cost_buffers()
    trim(device)
    torch.cuda.empty_cache()


def stats():
    with _LOCK:
        return dict(_STATS, idle_blocks=len(_IDLE), idle_bytes=sum(b for _, _, b in list(_IDLE.values())))


def _raw(nbytes, dev, keep):
    torch = nat.torch_mod()
    lib = nat.load()
    idx = dev.index
    stream = torch.cuda.current_stream(dev).cuda_stream
    with _LOCK:
        if idx not in _TOTAL:
            free_b, total_b = ctypes.c_size_t(), ctypes.c_size_t()
            nat.check(lib.pm_device_memory(idx, ctypes.byref(free_b), ctypes.byref(total_b)))
            _TOTAL[idx] = total_b.value
        hit = next((p for p, v in list(_IDLE.items()) if v == (idx, stream, nbytes)), None)
        if hit is not None:
            _IDLE.pop(hit, None)
            _STATS["reused"] += 1
            return _Block(hit, nbytes, idx, stream, keep)
    out = ctypes.c_void_p()
    rc = lib.pm_device_alloc(idx, nbytes, ctypes.byref(out))
    if rc == -5:                   # PM_ERR_NO_MEMORY: give back what nothing uses, once, and ask again
        release_everything_idle(dev)
        rc = lib.pm_device_alloc(idx, nbytes, ctypes.byref(out))
        if rc == -5:
            free_b, total_b = ctypes.c_size_t(), ctypes.c_size_t()
            lib.pm_device_memory(idx, ctypes.byref(free_b), ctypes.byref(total_b))
            raise torch.OutOfMemoryError("platymatch_amd: no block of %.2f GiB left on %s (%.2f GiB free of %.2f; idle blocks and kept cost "
                                         "buffers have been returned to the driver)" % (nbytes / 2.0 ** 30, dev, free_b.value / 2.0 ** 30, total_b.value / 2.0 ** 30))
    nat.check(rc)
    with _LOCK:
        _STATS["raw_allocations"] += 1
    return _Block(out.value, nbytes, idx, stream, keep)


def big_empty(shape, dtype, device, keep=True):
    """torch.empty(shape, dtype=dtype, device=device) for a matrix of the path.  From BIG_BLOCK_BYTES on the storage is a raw block
    (module docstring); keep=False: the block goes back to the driver as soon as its last view is dropped (the kept cost buffer,
    which cost_buffers.py holds on to itself), otherwise it waits idle for the next request of its size on this stream."""
    torch = nat.torch_mod()
    dev = _resolve(device)
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    need = int(np.prod(shape, dtype=np.int64)) * torch.empty(0, dtype=dtype).element_size()
    if dev.type != "cuda" or need < BIG_BLOCK_BYTES or getattr(_THREAD, "depth", 0) > 0:
        return torch.empty(shape, dtype=dtype, device=dev)
    nbytes = (need + _GRANULE - 1) // _GRANULE * _GRANULE
    block = _raw(nbytes, dev, keep)
    flat = torch.as_tensor(block, device=dev)          # uint8 [nbytes] on the block's memory; torch holds `block` until the last view dies
    if flat.data_ptr() != block.ptr:
        raise nat.NativeError("torch copied a raw device block instead of wrapping it")
    return flat[:need].view(dtype).view(shape)
