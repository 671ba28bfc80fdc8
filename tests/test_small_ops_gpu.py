"""Small training-step helpers of the C ABI against plain torch: t2s_zero_fill (the one fill of a step's accumulator arena) and
t2s_bn_running_update (nn.BatchNorm1d's training-mode bookkeeping, torch/nn/modules/batchnorm.py as used by the reference's
tacotron/modules.py:105-137 in .train() mode)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_bytes", [16, 4, 20, 4096 + 12, 3 * 1024 * 1024 + 4])
def test_zero_fill(n_bytes):
    from text2speech_amd import _lib
    dev = torch.device("cuda:0")
    buf = torch.full((n_bytes // 4 + 64,), 7.0, device=dev)
    _lib.call("t2s_zero_fill", _lib.ptr(buf), n_bytes, _lib.current_stream())
    torch.cuda.synchronize()
    n = n_bytes // 4
    assert float(buf[:n].abs().max()) == 0.0
    assert bool((buf[n:] == 7.0).all()), "wrote past the requested range"


@pytest.mark.parametrize("C,n,momentum", [(512, 32 * 800, 0.1), (80, 7, 0.1), (3, 1, 0.5)])
def test_bn_running_update_matches_torch(C, n, momentum):
    from text2speech_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + n)
    x = torch.randn(max(n, 2), C, generator=g)[:n].to(dev)          # n samples per channel
    bn = torch.nn.BatchNorm1d(C, momentum=momentum).to(dev).train()
    bn.running_mean.copy_(torch.randn(C, generator=g))
    bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    rm, rv, nb = bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()
    if n > 1:
        bn(x)                                                       # torch updates its running statistics
    mean, var = x.mean(0), x.var(0, unbiased=False) if n > 1 else torch.zeros(C, device=dev)
    _lib.call("t2s_bn_running_update", _lib.ptr(mean.contiguous()), _lib.ptr(var.contiguous()), _lib.ptr(rm), _lib.ptr(rv),
              _lib.ptr(nb), float(momentum), n, C, _lib.current_stream())
    torch.cuda.synchronize()
    assert int(nb) == 1
    if n > 1:
        assert torch.allclose(rm, bn.running_mean, rtol=1e-6, atol=1e-7)
        assert torch.allclose(rv, bn.running_var, rtol=1e-5, atol=1e-7)
    else:       # one sample: no unbiasing possible, the batch variance (0) enters as it is
        assert torch.allclose(rm, bn.running_mean * (1 - momentum) + momentum * mean, rtol=1e-6, atol=1e-7)
