"""One data-parallel rank of the WaveGlow training step (helper of tests/test_waveglow_dp_gpu.py, not a test).

Started as a fresh child process per rank (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment).  The ranks
share the box's one GPU and talk over gloo, so the production path - text2speech_amd.distributed.apply_gradient_allreduce
on this package's WaveGlow, GradSync.reduce_async called from inside glow_autograd.backward_train with the side stream in
play, FusedAdam after it - runs at world size 2 without a second GPU.  Reference semantics:
waveglow/distributed.py:100-129 (broadcast from rank 0; gradients = mean over ranks), waveglow/train.py:110-124.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, channels = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from text2speech_amd import _lib, synth
    from text2speech_amd import distributed as D
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    from text2speech_amd.optim import FusedAdam
    _lib.load()
    torch.cuda.set_device(0)
    backend = os.environ.get("T2S_DP_BACKEND", "gloo")
    D.init_distributed(rank, world, None, backend, "tcp://%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"]))
    cfg = dict(synth.WAVEGLOW_SMALL)
    cfg["WN_config"] = dict(cfg["WN_config"], n_channels=channels)
    # every rank starts from DIFFERENT weights: after apply_gradient_allreduce all must hold rank 0's
    sd = synth.waveglow_state(cfg, seed=1234 + 1000 * rank)
    m = WaveGlow(**cfg)
    m.load_state_dict(sd)
    m = m.to("cuda:0").train()
    same = D.apply_gradient_allreduce(m)
    assert same is m
    w0 = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    mel, audio = synth.waveglow_inputs(2, 2048, seed=50 + rank)           # each rank its own batch
    opt = FusedAdam(m.parameters(), lr=1e-3)
    m.zero_grad(set_to_none=True)
    loss = WaveGlowLoss(1.0)(m((mel.cuda(), audio.cuda())))
    loss.backward()
    torch.cuda.synchronize()
    sync = m._eng().grad_sync
    grads = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    red = D.reduce_tensor(loss.detach(), world)                            # distributed.py:37-41
    opt.step()
    torch.cuda.synchronize()
    w1 = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    torch.save(dict(w0=w0, grads=grads, w1=w1, loss=float(loss), loss_mean=float(red), n_buckets=sync.n_buckets,
                    pending=len(sync.pending), bytes=sync.bytes), os.path.join(out_dir, "rank%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
