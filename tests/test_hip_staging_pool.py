"""The page-locked staging pool of the sessions (engine.PinnedPool, round 4): buffers of exact size from hipHostMalloc that torch
sees as pinned, sized ONCE from the file-size hint, handed to the reader with the largest file first, and given back to the
driver when their last view is gone."""
import gc
import threading
import time

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from epilogos_amd import engine
    engine.require_gpu()
    return engine


def test_exact_size_pinned_buffer_and_async_copy(eng):
    n = (1 << 20) + 12345                                           # not a power of two: torch's own allocator would round it up
    t = eng._pinned_bytes(n)
    assert t.numel() == n and t.dtype == torch.int8 and t.is_pinned()
    v = t[:4096].view(64, 64)
    v.numpy()[:] = np.arange(4096, dtype=np.int8).reshape(64, 64)
    d = torch.empty((64, 64), dtype=torch.int8, device="cuda")
    d.copy_(v, non_blocking=True)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy(), np.arange(4096, dtype=np.int8).reshape(64, 64))
    del t, v, d
    gc.collect()                                                    # (the finalizer hands the memory back; nothing to assert but no crash)


def test_pool_sizes_its_buffers_once_from_the_hint(eng):
    pool = eng.PinnedPool(3, in_order=False)
    pool.hint({0: 1000, 1: 4000, 2: 2000})                          # bytes of the input files behind the tickets
    a = pool.acquire(0, 1 << 20)                                    # the smallest file completes first: 1 MiB of states
    assert a.numel() >= int((1 << 20) * 4 * 1.05) - 8               # ... sized for the largest file (4 x) + 5 %
    pool.release(a)
    b = pool.acquire(1, 4 << 20)                                    # the largest file fits the buffer that is already there
    assert b.data_ptr() == a.data_ptr() and b.numel() == a.numel()
    pool.release(b)


def test_pool_serves_the_largest_waiting_file_first(eng):
    pool = eng.PinnedPool(1, in_order=False)
    pool.hint({0: 10, 1: 1000, 2: 100})
    first = pool.acquire(0, 1 << 20)
    order = []

    def want(ticket):
        buf = pool.acquire(ticket, 1 << 20)
        order.append(ticket)
        time.sleep(0.05)
        pool.release(buf)
    threads = [threading.Thread(target=want, args=(t,)) for t in (2, 1)]
    threads[0].start()                                              # ticket 2 (100 bytes of input) waits first ...
    deadline = time.time() + 20
    while 2 not in pool.waiting and time.time() < deadline:
        time.sleep(0.01)
    threads[1].start()                                              # ... ticket 1 (1000 bytes) joins the queue
    while pool.waiting != {1, 2} and time.time() < deadline:
        time.sleep(0.01)
    assert pool.waiting == {1, 2}
    pool.release(first)
    for th in threads:
        th.join(timeout=10)
    assert order == [1, 2]                                          # the larger file overtook


def test_pool_abort_wakes_waiting_readers(eng):
    pool = eng.PinnedPool(1, in_order=False)
    held = pool.acquire(0, 1 << 20)
    err = []

    def want():
        try:
            pool.acquire(1, 1 << 20)
        except RuntimeError as e:
            err.append(str(e))
    th = threading.Thread(target=want)
    th.start()
    time.sleep(0.1)
    pool.abort()
    th.join(timeout=10)
    assert err and "aborted" in err[0]
    del held
