"""Adam with torch.optim.Adam semantics (reference waveglow/train.py:79,124) as ONE table-driven kernel
launch over every parameter (t2s_adam_table), instead of ~5 elementwise kernels per tensor."""
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}
        self.grad_scale = 1.0       # e.g. 1/world_size after an all-reduce(SUM)

    def _table(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        t = self._tables.get(gi)
        if t is not None and t["key"] == key:
            return t
        rows, blk = [], 0
        for p in ps:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                raise _lib.T2SError("FusedAdam needs contiguous float32 GPU parameters and gradients")
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blk])
            blk += -(-p.numel() // 1024)
        dev = ps[0].device
        t = dict(key=key, jobs=torch.tensor(rows, dtype=torch.int64).to(dev), n=len(rows), blocks=blk)
        self._tables[gi] = t
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group["params"]):
                continue
            t = self._table(gi, group)
            group["step"] = group.get("step", 0) + 1
            b1, b2 = group["betas"]
            _lib.call("t2s_adam_table", _lib.ptr(t["jobs"]), t["n"], t["blocks"], float(group["lr"]), float(b1), float(b2),
                      float(group["eps"]), int(group["step"]), float(self.grad_scale), float(group["weight_decay"]),
                      _lib.current_stream())
        return loss
