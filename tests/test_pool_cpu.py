"""Host logic of the training-step buffer pool (text2speech_amd/tacotron/tacotron.py: pool_take / _Lease): buffers are keyed by
call site + shape, come back when their holder is released, are never shared between two live holders, and `zero_once` buffers
are zero-filled only when created (or, with an `extent`, again when the valid extent changes: tests/test_tacotron_train_gpu.py).
The bound on the pool's size is in tests/test_host_logic_cpu.py.  Runs on CPU tensors."""
import gc

import torch

from text2speech_amd.tacotron.tacotron import BufferPool, pool_take


def test_lease_returns_buffer_when_holder_is_released():
    pool, h1 = BufferPool(), []
    a = pool_take(pool, h1, "site", (4, 8), torch.float32, "cpu")
    assert a.shape == (4, 8)
    h2 = []
    b = pool_take(pool, h2, "site", (4, 8), torch.float32, "cpu")
    assert b.data_ptr() != a.data_ptr(), "two live holders must not share a buffer"
    pa = a.data_ptr()
    del a
    h1.clear()
    gc.collect()
    c = pool_take(pool, h2, "site", (4, 8), torch.float32, "cpu")
    assert c.data_ptr() == pa, "a released buffer is reused by the next taker of the same call site and shape"


def test_keys_separate_tags_shapes_and_dtypes():
    pool, h = BufferPool(), []
    a = pool_take(pool, h, "x", (8,), torch.float32, "cpu")
    b = pool_take(pool, h, "y", (8,), torch.float32, "cpu")
    c = pool_take(pool, h, "x", (16,), torch.float32, "cpu")
    d = pool_take(pool, h, "x", (8,), torch.bfloat16, "cpu")
    assert len({t.data_ptr() for t in (a, b, c, d)}) == 4
    del a, b, c, d
    h.clear()
    gc.collect()
    assert len(pool.free) == 4


def test_zero_once_is_cleared_at_creation_only():
    pool, h = BufferPool(), []
    a = pool_take(pool, h, "planes", (3, 32), torch.float32, "cpu", zero_once=True)
    assert float(a.abs().max()) == 0.0
    a.fill_(5.0)                    # the owner writes its valid region ...
    del a
    h.clear()
    gc.collect()
    b = pool_take(pool, h, "planes", (3, 32), torch.float32, "cpu", zero_once=True)
    assert float(b.min()) == 5.0, "... and a reused buffer is handed back as it is (its padding depends on the shape only)"


def test_zero_once_with_extent_is_cleared_again_when_the_extent_changes():
    """Planes whose padded shape is shared by several valid lengths (Lp = ceil(T / 256) * 256 + 2 * halo): same extent -> handed back
    as it is, other extent -> cleared (the producers write rows below T only)."""
    pool, h = BufferPool(), []
    a = pool_take(pool, h, "planes", (4, 8), torch.float32, "cpu", zero_once=True, extent=(40,))
    a.fill_(7.0)
    del a
    h.clear()
    gc.collect()
    b = pool_take(pool, h, "planes", (4, 8), torch.float32, "cpu", zero_once=True, extent=(40,))
    assert float(b.min()) == 7.0
    del b
    h.clear()
    gc.collect()
    c = pool_take(pool, h, "planes", (4, 8), torch.float32, "cpu", zero_once=True, extent=(37,))
    assert float(c.abs().max()) == 0.0
