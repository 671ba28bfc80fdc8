"""Adam with torch.optim.Adam semantics (reference waveglow/train.py:79,124; train.py:187-189,225) as ONE table-driven
kernel launch over every parameter (t2s_adam_table), instead of ~5 elementwise kernels per tensor.

State layout is torch.optim.Adam's (per parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so an optimizer ``state_dict``
written by the reference's ``torch.optim.Adam`` (waveglow/train.py:41-50) loads here and continues with the right bias
correction, and one written here loads into ``torch.optim.Adam``.
"""
import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._tables = {}
        self._step_t = {}           # per group: one shared host tensor holding the step count (torch.optim.Adam's state layout)
        self.grad_scale = 1.0       # e.g. 1/world_size after an all-reduce(SUM)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}           # the cached job tables point at the moment buffers that were just replaced
        self._step_t = {}
        for group in self.param_groups:
            group.pop("step", None)     # re-derived from the loaded per-parameter state on the next step()

    def state_dict(self):
        """torch.optim.Adam's layout with a step tensor OF ITS OWN per parameter.  Internally every parameter of a group points at
        one shared host tensor (a step costs one fill_, not one per parameter); emitted as such, `torch.optim.Adam` loading the
        dict would advance the shared counter once per parameter per step."""
        sd = super().state_dict()
        for st in sd["state"].values():
            s = st.get("step")
            if s is not None:
                st["step"] = torch.tensor(float(s.item() if torch.is_tensor(s) else s), dtype=torch.float32)
        return sd

    @staticmethod
    def _loaded_step(group, state):
        """Step count of a group: torch.optim.Adam keeps it per parameter (tensor or int)."""
        best = 0
        for p in group["params"]:
            s = state.get(p, {}).get("step")
            if s is not None:
                best = max(best, int(s.item() if torch.is_tensor(s) else s))
        return best

    def _table(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                     self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        t = self._tables.get(gi)
        if t is not None and t["key"] == key:
            return t
        rows, blk = [], 0
        for p in ps:
            st = self.state[p]
            for x in (p, p.grad, st["exp_avg"], st["exp_avg_sq"]):
                if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous():
                    raise _lib.T2SError("FusedAdam needs contiguous float32 GPU parameters, gradients and moments")
            rows.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), blk])
            blk += -(-p.numel() // 1024)
        dev = ps[0].device
        t = dict(key=key, jobs=torch.tensor(rows, dtype=torch.int64).to(dev), n=len(rows), blocks=blk, params=ps)
        self._tables[gi] = t
        return t

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group["params"]):
                continue
            t = self._table(gi, group)
            if "step" not in group:
                group["step"] = self._loaded_step(group, self.state)
            if gi not in self._step_t:
                self._step_t[gi] = torch.zeros((), dtype=torch.float32)
            group["step"] += 1
            self._step_t[gi].fill_(group["step"])
            for p in t["params"]:
                self.state[p]["step"] = self._step_t[gi]
            b1, b2 = group["betas"]
            _lib.call("t2s_adam_table", _lib.ptr(t["jobs"]), t["n"], t["blocks"], float(group["lr"]), float(b1), float(b2),
                      float(group["eps"]), int(group["step"]), float(self.grad_scale), float(group["weight_decay"]),
                      _lib.current_stream())
            # The kernel wrote the parameters through raw pointers: tell autograd / every cache keyed on tensor versions
            # (WaveGlow's packed-weight cache, glow._Engine.pack_weights) that they changed.  No kernel is launched.
            torch.autograd.graph.increment_version(t["params"])
        return loss
