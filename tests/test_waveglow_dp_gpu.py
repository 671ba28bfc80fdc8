"""The data-parallel WaveGlow training path asserted at world size 2.

Two fresh child ranks share this box's one GPU and all-reduce over gloo (tests/dp_rank_worker.py); the code under test is
the production path: apply_gradient_allreduce on this package's WaveGlow, 13 flat buckets handed to the process group from
inside the hand-written backward.  Checked against single-process runs in THIS process:
  * weights after apply_gradient_allreduce == rank 0's initial weights on every rank (distributed.py:100-103)
  * every parameter gradient on every rank == mean of the two ranks' single-process gradients (distributed.py:105-129)
  * the Adam step that follows leaves both ranks with identical weights (waveglow/train.py:124)
"""
import os
import socket
import subprocess
import sys

import pytest
import torch

from text2speech_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("channels", [64, 256])
def test_world2_gradients_are_the_mean_and_weights_broadcast(tmp_path, channels):
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    _lib.load()
    world = 2
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_rank_worker.py"), str(tmp_path), str(channels)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    res = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]

    cfg = dict(synth.WAVEGLOW_SMALL)
    cfg["WN_config"] = dict(cfg["WN_config"], n_channels=channels)
    sd0 = synth.waveglow_state(cfg, seed=1234)
    # (1) broadcast: every rank holds rank 0's weights
    for r in range(world):
        for n, w in res[r]["w0"].items():
            assert torch.equal(w, sd0[n]), (r, n)
        assert res[r]["n_buckets"] == cfg["n_flows"] + 1 and res[r]["pending"] == 0
    # (2) single-process gradients of each rank's batch, in this process
    local, losses = [], []
    for r in range(world):
        m = WaveGlow(**cfg)
        m.load_state_dict(sd0)
        m = m.to(DEV).train()
        mel, audio = synth.waveglow_inputs(2, 2048, seed=50 + r)
        loss = WaveGlowLoss(1.0)(m((mel.to(DEV), audio.to(DEV))))
        loss.backward()
        torch.cuda.synchronize()
        local.append({n: p.grad.detach().cpu() for n, p in m.named_parameters()})
        losses.append(float(loss))
    for r in range(world):
        assert abs(res[r]["loss"] - losses[r]) < 1e-6
        assert abs(res[r]["loss_mean"] - sum(losses) / world) < 1e-6
    worst = 0.0
    for n in local[0]:
        want = (local[0][n] + local[1][n]) * 0.5
        for r in range(world):
            got = res[r]["grads"][n]
            assert got.shape == want.shape, n
            e = _rel(got, want)
            worst = max(worst, e)
            assert e < 1e-6, (n, r, e)
        assert torch.equal(res[0]["grads"][n], res[1]["grads"][n]), n          # both ranks hold the SAME averaged gradient
    # the gradients differ between the two batches, so the mean is not either of them
    assert _rel(local[0]["WN.3.in_layers.2.weight_v"], local[1]["WN.3.in_layers.2.weight_v"]) > 1e-2
    # (3) after the optimizer step both ranks still agree, and moved
    for n in res[0]["w1"]:
        assert torch.equal(res[0]["w1"][n], res[1]["w1"][n]), n
    assert not torch.equal(res[0]["w1"]["WN.0.start.bias"], res[0]["w0"]["WN.0.start.bias"])
