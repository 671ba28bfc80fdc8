"""Host-side logic that needs no GPU: the bounded buffer pool of the Tacotron training step and bench.py's watchdog."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_buffer_pool_is_bounded_and_lru():
    from text2speech_amd.tacotron.tacotron import BufferPool, pool_take
    pool = BufferPool(cap_bytes=10 * 4096)
    holder = []
    a = pool_take(pool, holder, "a", (1024,), torch.float32, "cpu")           # 4 KiB each
    b = pool_take(pool, holder, "b", (1024,), torch.float32, "cpu")
    assert a.data_ptr() != b.data_ptr() and pool.free_bytes == 0             # leased buffers are in no free list
    del a, b
    holder.clear()                                                           # leases released: both go back
    assert pool.free_bytes == 2 * 4096
    a2 = pool_take(pool, holder, "a", (1024,), torch.float32, "cpu")
    assert pool.free_bytes == 4096                                           # reused, not allocated
    holder.clear()
    del a2
    # twenty distinct shapes (a ragged epoch): the pool never holds more than its cap, oldest keys go first
    for i in range(20):
        h = []
        pool_take(pool, h, "s", (1024 + i,), torch.float32, "cpu")
        h.clear()
        assert pool.free_bytes <= pool.cap_bytes
    assert pool.evicted > 0
    assert ("a", (1024,), torch.float32, "cpu") not in pool.free            # least recently used: gone
    assert ("s", (1043,), torch.float32, "cpu") in pool.free                # most recent: kept
    # a single buffer larger than the cap is not kept at all
    h = []
    pool_take(pool, h, "big", (1 << 20,), torch.float32, "cpu")
    h.clear()
    assert pool.free_bytes <= pool.cap_bytes


def test_buffer_pool_zero_once_shape_only_keeps_contents():
    """zero_once without an extent: zeroed at creation only (shape-determined padding)."""
    from text2speech_amd.tacotron.tacotron import BufferPool, pool_take
    pool = BufferPool()
    h = []
    t = pool_take(pool, h, "w", (8,), torch.float32, "cpu", zero_once=True)
    assert float(t.abs().sum()) == 0.0
    t.fill_(3.0)
    del t
    h.clear()
    t = pool_take(pool, h, "w", (8,), torch.float32, "cpu", zero_once=True)
    assert float(t.sum()) == 24.0


def test_bench_watchdog_exits_nonzero():
    """a bench process that gives up on its train block must print the headline line WITH the error and leave with a
    non-zero status.  The watchdog function is driven in a child process (it ends in os._exit)."""
    code = (
        "import sys, json; sys.path.insert(0, %r); import bench\n"
        "out = {'metric': 'm', 'value': 1.0}\n"
        "bench.watchdog_fired(out, 'waveglow_train_dp', 0, 240)\n"
        "print('not reached')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and "not reached" not in r.stdout
    out = json.loads(lines[0])
    assert out["value"] == 1.0 and "timeout" in out["waveglow_train_dp"]["error"] and "watchdog" in out["error"]


def test_bench_watchdog_timer_path():
    """The same through a real threading.Timer while the main thread hangs in a 'collective' (a sleep)."""
    code = (
        "import sys, threading, time; sys.path.insert(0, %r); import bench\n"
        "t = threading.Timer(0.2, lambda: bench.watchdog_fired({'value': 2.0}, 'waveglow_train', 0, 0)); t.daemon = True; t.start()\n"
        "time.sleep(30)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    assert json.loads(r.stdout.strip())["waveglow_train"]["error"].startswith("timeout")


def test_avg_op_follows_the_groups_backend():
    import torch.distributed as dist
    from text2speech_amd import distributed as D
    assert D._avg_op("nccl") == dist.ReduceOp.AVG
    assert D._avg_op("gloo") == dist.ReduceOp.SUM
