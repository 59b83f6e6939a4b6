"""Optimizer for param group "fields": torch.optim.Adam semantics (AdamOptimizerConfig(lr=2e-2, eps=1e-15),
umhs_config.py:59-64) executed by ONE fused HIP launch over the flat parameter buffer, with the
``clamp_endmembers`` callback (umhs_model.py:568-572) folded into the same pass, and the data-parallel gradient
reduction (the reference's DDP wrap, umhs_pipeline.py:110-113) as one RCCL all-reduce of the flat gradient."""
from __future__ import annotations

import math
import os
from typing import Optional, Tuple

import torch

from . import ops
from .parallel import allreduce_flat_grad, world


def exp_decay_lr(step: int, lr_init: float = 2e-2, lr_final: float = 1e-5, max_steps: int = 30000) -> float:
    """nerfstudio ExponentialDecayScheduler (no warm-up), umhs_config.py:63."""
    t = min(max(step / max_steps, 0.0), 1.0)
    return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class UMHSAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 2e-2, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-15,
                 clamp_range: Tuple[int, int] = (0, 0), lr_final: Optional[float] = None, max_steps: int = 30000, weight_decay: float = 0.0):
        if weight_decay:  # accepted because nerfstudio's AdamOptimizerConfig.setup() passes it; the reference leaves it at 0
            raise NotImplementedError("UMHSAdam has no weight decay (umhs_config.py:61 uses AdamOptimizerConfig(lr=2e-2, eps=1e-15))")
        defaults = dict(lr=lr, betas=betas, eps=eps, clamp_range=clamp_range, lr_init=lr, lr_final=lr_final, max_steps=max_steps)
        super().__init__(params, defaults)

    def _hyper(self, group, st):
        """(learning rate, step number) of the update that follows ``st["step"]`` updates."""
        step = st["step"] + 1
        lr = group["lr"]
        if group["lr_final"] is not None:
            lr = exp_decay_lr(step - 1, group["lr_init"], group["lr_final"], group["max_steps"])
        return lr, step

    @torch.no_grad()
    def arm_fused(self) -> bool:
        """Called by a single-GPU trainer right before a backward that ``step()`` will follow: hands the flat parameter's gradient
        sink the hyper-parameters and moment buffers of the coming update, so that the backward's last kernel (the bucket reduce
        of the hash-grid gradient, LDS-bound) also applies Adam to the dense levels of the hash table (HBM-bound: it hides inside,
        and the gradient is not read back).  ``step()`` then skips that range.  Same arithmetic, same bits (tested)."""
        if os.environ.get("UMHS_FUSED_ADAM", "1") == "0" or world()[1] != 1:
            return False
        armed = False
        for group in self.param_groups:
            for p in group["params"]:
                sink = getattr(p, "_umhs_grad_sink", None)
                sparse = getattr(p, "_umhs_live_rows", None)
                if sink is None or not p.is_cuda or p.grad is not None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                lr, step = self._hyper(group, st)
                sink.adam_done = None
                sink.fused_adam = dict(lr=lr, betas=group["betas"], eps=group["eps"], step=step, exp_avg=st["exp_avg"],
                                       exp_avg_sq=st["exp_avg_sq"], level_begin=sink.sparse_levels if sparse is not None else 0)
                armed = True
        return armed

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("UMHSAdam works on the flat fp32 'fields' buffer")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                lr, st["step"] = self._hyper(group, st)
                sink = getattr(p, "_umhs_grad_sink", None)
                done_by_backward = None  # (begin, end) elements the backward already updated (arm_fused)
                if sink is not None:
                    sink.fused_adam, ad, sink.adam_done = None, sink.adam_done, None
                    if ad is not None:
                        if ad[0] != st["step"]:
                            raise RuntimeError("a backward applied a fused Adam step that does not belong to this optimizer step")
                        done_by_backward = ad[1:]
                cb, ce = group["clamp_range"]
                sparse = getattr(p, "_umhs_live_rows", None)  # (rows int64 [n], end): elements [0, end) hold sparse hash levels

                def update(a: int, b: int, scale: float) -> None:
                    """Adam on elements [a, b) of the flat buffers; the sparse coarse levels only on their live rows."""
                    if sparse is not None and a == 0 and b >= sparse[1] and sparse[0].device == p.device:
                        if done_by_backward is not None and done_by_backward[0] <= sparse[1] and done_by_backward[1] < b:
                            # the backward took everything between the sparse rows and the MLP tail: rows + tail in one launch
                            ops.adam_step_rows_range(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], sparse[0], done_by_backward[1], b,
                                                     st["step"], lr, group["betas"], group["eps"], grad_scale=scale, clamp_range=(cb, ce))
                            return
                        ops.adam_step_rows(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], sparse[0], st["step"], lr, group["betas"],
                                           group["eps"], grad_scale=scale)
                        a = sparse[1]
                        if a == b:
                            return
                    pieces = [(a, b)]
                    if done_by_backward is not None:  # skip what the reduce pass of the backward has updated
                        da, db = done_by_backward
                        pieces = [(x, y) for x, y in ((a, min(b, da)), (max(a, db), b)) if x < y]
                    for x, y in pieces:
                        clamp = (max(cb, x) - x, min(ce, y) - x) if (cb < y and ce > x) else (0, 0)
                        ops.adam_step(p.data[x:y], p.grad[x:y], st["exp_avg"][x:y], st["exp_avg_sq"][x:y], st["step"], lr, group["betas"],
                                      group["eps"], grad_scale=scale, clamp_range=clamp)

                done = 0
                if sink is not None:
                    # Segments were all-reduced while the backward was still running: update each one as soon as ITS reduction
                    # has landed, so the Adam pass of segment k hides under the transfer of segments k+1..
                    for a, b in sink.reduced_segments(p.grad):
                        update(a, b, 1.0 / world()[1])
                        done += b - a
                if done == 0:
                    update(0, p.numel(), allreduce_flat_grad(p.grad))  # one RCCL all-reduce of the whole gradient (no-op at world 1)
