"""One training step as ONE hipGraph launch (SURVEY.md §8(f) N1 + N2 end to end).

The reference's loop body (train.py:197-210) is: collate the dense batch on the host, `.cuda()`, forward, loss,
backward, clip_grad_norm, optimizer.step — several hundred launches and 20 MB of PCIe per DD step.  Here the step is
captured once:

    dp_build_batch_packed (edge lists read from pinned host memory by the kernel itself: no copy node)
    -> dp_gather_labels -> dp_encoder_forward_packed -> dp_loss_forward -> dp_encoder_backward_packed
    -> dp_clip_adam_step_counted (Adam's step count lives on the device)

and a step is: write the batch's edge lists into a pinned slot, `graph.replay()`.  Two slots / two graphs alternate, so
the host prepares step i + 1 while the GPU runs step i.

    step = CapturedTrainStep(model, FusedClipAdam(model, lr=1e-3, clip=2.0, device_step_counter=True), builder, 20)
    for idx in batches:
        loss = step(idx)          # device scalar of the slot; read it (loss.item()) only when you must
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import _lib
from .batch_builder import DeviceBatchBuilder
from .optim import FusedClipAdam


class _Slot:
    def __init__(self, B, cap_e, cap_n):
        ints = 2 * cap_e + 2 * (B + 1) + cap_n
        ints += ints & 1                                  # the int64 labels start 8-byte aligned
        self.buf = torch.zeros(ints + 2 * B, dtype=torch.int32).pin_memory()
        v = self.buf.numpy()
        o = 0
        self.src = v[o:o + cap_e]; o += cap_e
        self.dst = v[o:o + cap_e]; o += cap_e
        self.edge_ptr = v[o:o + B + 1]; o += B + 1
        self.lab = v[o:o + cap_n]; o += cap_n
        self.node_ptr = v[o:o + B + 1]; o += B + 1
        o += o & 1
        self.label = v[o:o + 2 * B].view(np.int64)
        base = self.buf.data_ptr()
        addr = lambda a: base + (a.__array_interface__["data"][0] - v.__array_interface__["data"][0])   # noqa: E731
        self.ptrs = [addr(self.src), addr(self.dst), addr(self.edge_ptr), addr(self.lab), addr(self.node_ptr)]
        self.label_ptr = addr(self.label)
        self.event = torch.cuda.Event()
        self.graph = None
        self.loss = None
        self.errors = None


class CapturedTrainStep:
    """build -> forward -> loss -> backward -> clip + Adam of a SoftPoolingGcnEncoder (linkpred=False) as one graph
    launch per step.  `optimizer` must be a FusedClipAdam(device_step_counter=True); every batch has `batch_size`
    graphs (the reference's DataLoader drops nothing but its last batch may be short: run that one eagerly)."""

    def __init__(self, model, optimizer: FusedClipAdam, builder: DeviceBatchBuilder, batch_size: int, slots: int = 2):
        if not optimizer.device_step_counter:
            raise ValueError("CapturedTrainStep needs FusedClipAdam(device_step_counter=True)")
        if getattr(model, "linkpred", False):
            raise ValueError("the packed-adjacency step has no link-prediction loss (it reads the fp32 adjacency)")
        self.model, self.opt, self.builder, self.B = model, optimizer, builder, int(batch_size)
        ds = builder.ds
        ecount = np.diff(ds.edge_ptr)
        ncount = np.diff(ds.node_ptr)
        top = lambda a: int(np.sort(a)[::-1][:self.B].sum())                       # noqa: E731
        self.cap_e, self.cap_n = max(top(ecount), 1), max(top(ncount), 1)
        self.max_edges = int(ecount.max()) if len(ecount) else 1
        self.slots = [_Slot(self.B, self.cap_e, self.cap_n) for _ in range(slots)]
        self.turn = 0
        self._capture()

    # -- host side of a step: the batch's edge lists, straight into the slot's pinned arrays
    def _fill(self, slot: _Slot, indices: Sequence[int]):
        ds, B = self.builder.ds, self.B
        if len(indices) != B:
            raise ValueError(f"captured step of {B} graphs got a batch of {len(indices)}")
        eo = no = 0
        for i, g in enumerate(indices):
            e0, e1 = int(ds.edge_ptr[g]), int(ds.edge_ptr[g + 1])
            n0, n1 = int(ds.node_ptr[g]), int(ds.node_ptr[g + 1])
            if n1 - n0 > self.builder.N:
                raise ValueError(f"graph {g} has {n1 - n0} nodes > max_nodes={self.builder.N}")
            slot.edge_ptr[i], slot.node_ptr[i] = eo, no
            slot.src[eo:eo + e1 - e0] = ds.edge_src[e0:e1]
            slot.dst[eo:eo + e1 - e0] = ds.edge_dst[e0:e1]
            slot.lab[no:no + n1 - n0] = ds.node_label[n0:n1]
            eo += e1 - e0
            no += n1 - n0
            slot.label[i] = ds.graph_label[g]
        slot.edge_ptr[B], slot.node_ptr[B] = eo, no

    def _body(self, slot: _Slot):
        lib = _lib.load()
        batch = self.builder.emit(slot.ptrs, self.B, self.max_edges, packed=True)
        label = torch.empty(self.B, device=self.builder.device, dtype=torch.int64)
        _lib.check(lib.dp_gather_labels(slot.label_ptr, label.data_ptr(), self.B, _lib.current_stream()),
                   "dp_gather_labels")
        self.model.zero_grad(set_to_none=True)
        ypred = self.model(batch["feats"], batch["adj"], batch["num_nodes_device"], assign_x=batch["assign_feats"])
        loss = self.model.loss(ypred, label)
        loss.backward()
        self.opt.step()
        return loss.detach(), batch["errors"]

    def _capture(self):
        model, opt = self.model, self.opt
        dev = self.builder.device
        first = list(range(self.B))
        for s in self.slots:
            self._fill(s, first)
        # warm-up runs real updates: snapshot what they change and put it back after the capture
        model._ensure_flat(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        opt.ensure_state()
        keep = (model._flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.step_dev.clone(), opt.step_count)
        with torch.cuda.stream(side):
            for _ in range(3):
                self._body(self.slots[0])
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        model.zero_grad(set_to_none=True)
        for s in self.slots:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                s.loss, s.errors = self._body(s)
            s.graph = g
            model.zero_grad(set_to_none=True)
        torch.cuda.synchronize(dev)
        # undo the warm-up
        model._flat.copy_(keep[0]); opt.exp_avg.copy_(keep[1]); opt.exp_avg_sq.copy_(keep[2]); opt.step_dev.copy_(keep[3])
        opt.step_count = keep[4]

    def __call__(self, indices: Sequence[int]) -> torch.Tensor:
        slot = self.slots[self.turn % len(self.slots)]
        self.turn += 1
        slot.event.synchronize()                          # the replay that last read this slot is done
        self._fill(slot, indices)
        slot.graph.replay()
        slot.event.record(torch.cuda.current_stream(self.builder.device))
        self.opt.step_count += 1
        return slot.loss

    def skipped_entries(self) -> int:
        """Out-of-range edge / label entries the builder skipped in the most recent step of each slot (0 = clean)."""
        return int(sum(int(s.errors.item()) for s in self.slots if s.errors is not None))
