"""Tacotron2Loss with the reference's interface (tacotron/loss_function.py:3-18; call site train.py:219-221):
``criterion(model_output, (mel_target, gate_target))`` -> scalar.  One HIP launch pair computes the three means and the three
gradients (csrc/loss_ops.hip); the backward only scales them by the upstream gradient.  No CPU path."""
import torch
from torch import nn

from .. import _lib


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mel_out, mel_post, gate_out, mel_target, gate_target):
        for t in (mel_out, mel_post, gate_out, mel_target, gate_target):
            if not t.is_cuda:
                raise _lib.T2SError("Tacotron2Loss (MI355X build) needs tensors in HBM; there is no CPU path")
        f = lambda t: t.detach().to(torch.float32).contiguous()
        mel, post, gate, tgt, gt = f(mel_out), f(mel_post), f(gate_out), f(mel_target), f(gate_target)
        if mel.shape != tgt.shape or post.shape != tgt.shape or gate.numel() != gt.numel():
            raise ValueError("Tacotron2Loss: shape mismatch between outputs and targets")
        dev = mel.device
        need = [mel_out.requires_grad, mel_post.requires_grad, gate_out.requires_grad]
        d_mel = torch.empty_like(mel) if need[0] else None
        d_post = torch.empty_like(post) if need[1] else None
        d_gate = torch.empty_like(gate) if need[2] else None
        partial = torch.empty(256 * 3, dtype=torch.float64, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        p = _lib.ptr
        _lib.call("t2s_taco_loss", p(mel), p(post), p(tgt), mel.numel(), p(gate), p(gt), gate.numel(), p(d_mel), p(d_post),
                  p(d_gate), p(partial), p(out), _lib.current_stream())
        ctx.grads = (d_mel, d_post, d_gate)
        ctx.shapes = (mel_out.shape, mel_post.shape, gate_out.shape)
        ctx.parts = out
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        outs = []
        for d, shp in zip(ctx.grads, ctx.shapes):
            outs.append(None if d is None else (d * g).view(shp))
        return outs[0], outs[1], outs[2], None, None


class Tacotron2Loss(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, model_output, targets):
        mel_target, gate_target = targets[0], targets[1]
        mel_out, mel_out_postnet, gate_out = model_output[0], model_output[1], model_output[2]
        return _LossFn.apply(mel_out, mel_out_postnet, gate_out.reshape(-1, 1), mel_target.detach(),
                             gate_target.detach().reshape(-1, 1))
