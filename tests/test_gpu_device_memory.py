"""The matrices' device memory (platymatch_amd/device_memory.py over pm_device_alloc / pm_device_free): blocks from 4 GiB come
straight from the driver, are never split, wait idle for a request of exactly their size and go back whole — so that a process
registering LARGE clouds of VARYING size does not strand its HBM in a caching allocator's split blocks (round 5: a third of such
registrations ran out of memory with 140 GB "reserved but unallocated")."""
import gc

import numpy as np
import pytest

from conftest import synth_pair

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dm():
    import torch
    from platymatch_amd import _native as nat, device_memory as D, pipeline as P
    from platymatch_amd.estimate_transform import perform_icp as pi
    nat.load()
    pi.VERBOSE = False
    P.release_cost_buffers()
    D.trim()
    gc.collect()
    torch.cuda.empty_cache()
    return D


def _driver_free():
    import torch
    return torch.cuda.mem_get_info()[0]


def test_small_requests_stay_with_torch_and_big_ones_are_raw_blocks(dm):
    import torch
    before = dm.stats()
    small = dm.big_empty((1000, 1000), torch.float64, "cuda")
    assert dm.stats().get("raw_allocations", 0) == before.get("raw_allocations", 0)
    free0 = _driver_free()
    big = dm.big_empty((5, 1 << 27), torch.float64, "cuda")                      # 5 GiB
    assert dm.stats()["raw_allocations"] == before.get("raw_allocations", 0) + 1
    assert free0 - _driver_free() >= 5 << 30                                     # the driver handed it out now, not torch's cache
    assert big.shape == (5, 1 << 27) and big.dtype == torch.float64 and big.is_contiguous() and big.data_ptr() % 256 == 0
    big.fill_(1.5)
    assert float(big[4, -1]) == 1.5 and float(big.sum()) == 1.5 * 5 * (1 << 27)
    view = big[2]                                                                # a view keeps the block alive
    ptr = big.data_ptr()
    del big
    gc.collect()
    assert dm.idle_bytes() == 0 and float(view[7]) == 1.5
    del view, small
    gc.collect()
    assert dm.idle_bytes() == 5 << 30                                            # idle, not yet back with the driver
    again = dm.big_empty((5, 1 << 27), torch.float64, "cuda")                    # the same size on the same stream: the same block
    assert again.data_ptr() == ptr and dm.idle_bytes() == 0 and dm.stats()["reused"] >= 1
    other = dm.big_empty((1 << 30, 5), torch.float32, "cuda")                    # another size: a block of its own
    assert other.data_ptr() != ptr
    del again, other
    gc.collect()
    assert dm.idle_bytes() == (5 << 30) + 5 * (4 << 30)
    free1 = _driver_free()
    assert dm.trim() == (5 << 30) + 5 * (4 << 30)
    assert _driver_free() - free1 >= 24 << 30 and dm.idle_bytes() == 0


def test_idle_blocks_are_bounded_and_unkept_blocks_go_back_at_once(dm, monkeypatch):
    import torch
    total = torch.cuda.mem_get_info()[1]
    monkeypatch.setattr(dm, "MAX_IDLE_FRACTION", (10 << 30) / total)             # room for two idle 4 GiB blocks, not three
    blocks = [dm.big_empty(((4 << 30) + k * (2 << 20),), torch.uint8, "cuda") for k in range(3)]
    ptrs = [b.data_ptr() for b in blocks]
    while blocks:
        blocks.pop(0)
        gc.collect()
    assert dm.stats()["idle_blocks"] == 2                                        # the oldest went back to the driver
    assert dm.big_empty(((4 << 30) + 2 * (2 << 20),), torch.uint8, "cuda").data_ptr() == ptrs[2]
    dm.trim()
    free0 = _driver_free()
    t = dm.big_empty((6 << 30,), torch.uint8, "cuda", keep=False)
    assert free0 - _driver_free() >= 6 << 30
    del t
    gc.collect()
    assert dm.idle_bytes() == 0 and _driver_free() >= free0 - (64 << 20)


def test_an_allocation_that_fails_returns_idle_memory_and_tries_again(dm, monkeypatch):
    import torch
    from platymatch_amd import pipeline as P
    monkeypatch.setattr(dm, "MAX_IDLE_FRACTION", 1.0)                            # (so that the large idle block below does stay idle)
    free, total = torch.cuda.mem_get_info()
    hog = dm.big_empty((int(0.55 * free),), torch.uint8, "cuda")
    del hog
    gc.collect()
    assert dm.idle_bytes() >= int(0.55 * free)
    kept = P.reserve(30000, 30000, cost_mode='exact')                            # + 57.6 GB kept, no registration holds it
    second = dm.big_empty((int(0.6 * free),), torch.uint8, "cuda")               # does not fit beside both: they are given back, then it does
    assert second.numel() == int(0.6 * free)
    assert dm.idle_bytes() == 0 and (kept == 0 or P.kept_cost_bytes(torch.device("cuda", torch.cuda.current_device())) == 0)
    with pytest.raises(torch.OutOfMemoryError):
        dm.big_empty((int(0.6 * free),), torch.uint8, "cuda")                    # while `second` is alive nothing can help
    del second
    gc.collect()
    dm.trim()


def test_large_registrations_of_varying_size_do_not_run_out_of_memory(dm):
    """The sequence that failed: exact and default mode alternating over pairs of 30 000-50 000 nuclei, N > M among them (transposed
    copies of 8 N M bytes, four in flight) — every call must succeed, both modes must agree, and afterwards the package must be able
    to hand (almost) all of the device back."""
    import torch
    from platymatch_amd import pipeline as P
    free_start = _driver_free()
    sizes = [(34000, 34000), (47000, 47000), (50200, 47000), (31500, 30900), (48300, 44500), (37200, 34500), (41000, 50100), (44700, 45700)]
    for k, (n, m) in enumerate(sizes):
        mv, fx, _ = synth_pair(max(n, m), 100 + k)
        mv, fx = np.ascontiguousarray(mv[:, :n]), np.ascontiguousarray(fx[:, :m])
        want = P.assignments(mv, fx, cost_mode='exact')
        got = P.assignments(mv, fx, cost_mode='auto')
        assert all(np.array_equal(want[h][0], got[h][0]) and np.array_equal(want[h][1], got[h][1]) for h in range(8)), (n, m)
    reserved_not_used = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
    assert reserved_not_used < 16 << 30, "torch's allocator holds %.1f GB it does not use" % (reserved_not_used / 1e9)
    import platymatch_amd
    platymatch_amd.release_memory()
    gc.collect()
    assert _driver_free() >= free_start - (8 << 30)
