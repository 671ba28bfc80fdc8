"""Host-side helpers for the split-bf16 "plane" layout (text2speech_amd/csrc/t2s_common.h).

Pure tensor reshuffling (plumbing): used by tests and by host code that has to hand an
ordinary [B, C, L] f32 tensor to a kernel that consumes planes.
"""
import torch

from . import _lib


def split_bf16(x):
    """x (f32) -> (hi, lo) bf16 with hi = bf16(x), lo = bf16(x - hi)."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.to(torch.float32)).to(torch.bfloat16)
    return hi, lo


def to_planes(x, halo, Lp=None):
    """[B, C, L] f32 -> (hi, lo) planes [B, ceil(C/32), Lp, 32] bf16, data rows at [halo, halo+L)."""
    B, C, L = x.shape
    if Lp is None:
        Lp = _lib.plane_rows(L, halo)
    nc = -(-C // 32)
    xp = torch.zeros(B, nc * 32, L, dtype=torch.float32, device=x.device)
    xp[:, :C] = x
    xp = xp.view(B, nc, 32, L).permute(0, 1, 3, 2)            # [B, nc, L, 32]
    hi, lo = split_bf16(xp)
    ph = torch.zeros(B, nc, Lp, 32, dtype=torch.bfloat16, device=x.device)
    pl = torch.zeros_like(ph)
    ph[:, :, halo:halo + L] = hi
    pl[:, :, halo:halo + L] = lo
    return ph, pl


def from_planes(ph, pl, C, L, halo):
    """(hi, lo) planes -> [B, C, L] f32."""
    B, nc = ph.shape[:2]
    v = ph[:, :, halo:halo + L].to(torch.float32) + pl[:, :, halo:halo + L].to(torch.float32)
    return v.permute(0, 1, 3, 2).reshape(B, nc * 32, L)[:, :C].contiguous()


def from_f32_planes(p, C, L, halo):
    B, nc = p.shape[:2]
    return p[:, :, halo:halo + L].permute(0, 1, 3, 2).reshape(B, nc * 32, L)[:, :C].contiguous()
