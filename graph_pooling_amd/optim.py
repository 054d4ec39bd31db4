"""Fused gradient clip + Adam over the flat parameter buffer.

Replaces the two lines every training step of the reference ends with (train.py:209-210, optimizer of :173):

    nn.utils.clip_grad_norm(model.parameters(), args.clip)
    optimizer.step()

by one C-ABI call (two kernel launches) on the encoder's flat fp32 parameter / gradient buffers — the same
buffers the kernels index and the data-parallel wrapper all-reduces.  Same arithmetic as
`torch.optim.Adam(lr, betas, eps)` (no weight decay, no amsgrad — the reference uses neither) after
`clip_grad_norm_(max_norm, norm_type=2)`.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib


class FusedClipAdam:
    """opt = FusedClipAdam(model, lr=1e-3, clip=2.0);  loss.backward();  opt.step()"""

    def __init__(self, model, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 clip: Optional[float] = None, device_step_counter: bool = False):
        """device_step_counter: keep Adam's step count in a device int that the update kernel increments and reads
        (dp_clip_adam_step_counted) instead of passing it from the host — nothing about the call changes from step to
        step, so `step()` can be captured in a hipGraph (train_step.CapturedTrainStep)."""
        self.model = getattr(model, "model", model)          # accept a DataParallelEncoder wrapper too
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.clip = float(clip) if clip is not None else 0.0
        self.step_count = 0
        self.device_step_counter = bool(device_step_counter)
        self.step_dev = None
        self.exp_avg = None
        self.exp_avg_sq = None
        self.total_norm = None
        self._ws = None

    def zero_grad(self, set_to_none: bool = True):
        self.model.zero_grad(set_to_none=set_to_none)

    def _flat_grad(self):
        m = self.model
        g = getattr(m, "_last_flat_grad", None)
        if g is not None:
            base = g.data_ptr()
            if all(p.grad is not None and p.grad.data_ptr() == base + 4 * off
                   for p, (off, _, _) in zip(m._flat_params, m._flat_index)):
                return g, True
        flat = torch.zeros(m._flat.numel(), device=m._flat.device, dtype=torch.float32)
        for p, (off, numel, _) in zip(m._flat_params, m._flat_index):
            if p.grad is not None:
                flat[off:off + numel].copy_(p.grad.reshape(-1))
        return flat, False

    def ensure_state(self):
        """Allocate the moments / step counter (zeros) for the model's current flat buffer; returns (n, device)."""
        lib = _lib.load()
        m = self.model
        device = next(m.parameters()).device
        _lib.require_gpu_tensor(next(m.parameters()), "model parameters")
        m._ensure_flat(device)
        n = m._flat.numel()
        if self.exp_avg is None or self.exp_avg.numel() != n or self.exp_avg.device != device:
            self.exp_avg = torch.zeros(n, device=device, dtype=torch.float32)
            self.exp_avg_sq = torch.zeros(n, device=device, dtype=torch.float32)
            self.total_norm = torch.zeros(1, device=device, dtype=torch.float32)
            self._ws = torch.empty(lib.dp_clip_adam_workspace_bytes(), device=device, dtype=torch.uint8)
            self.step_dev = torch.full((1,), self.step_count, device=device, dtype=torch.int32)
        return n, device

    @torch.no_grad()
    def step(self):
        lib = _lib.load()
        m = self.model
        n, device = self.ensure_state()
        grads, aliased = self._flat_grad()
        self.step_count += 1
        if self.device_step_counter:
            _lib.check(lib.dp_clip_adam_step_counted(m._flat.data_ptr(), grads.data_ptr(), self.exp_avg.data_ptr(),
                                                     self.exp_avg_sq.data_ptr(), n, self.step_dev.data_ptr(), self.lr,
                                                     self.betas[0], self.betas[1], self.eps, self.clip,
                                                     self.total_norm.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                     _lib.current_stream()), "dp_clip_adam_step_counted")
        else:
            _lib.check(lib.dp_clip_adam_step(m._flat.data_ptr(), grads.data_ptr(), self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), n, self.step_count, self.lr, self.betas[0],
                                             self.betas[1], self.eps, self.clip, self.total_norm.data_ptr(),
                                             self._ws.data_ptr(), self._ws.numel(), _lib.current_stream()),
                       "dp_clip_adam_step")
        if not aliased:           # keep .grad consistent with what clip_grad_norm_ would have left there
            for p, (off, numel, shape) in zip(m._flat_params, m._flat_index):
                if p.grad is not None:
                    p.grad.copy_(grads[off:off + numel].view(shape))
        return self.total_norm
