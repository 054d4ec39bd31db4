"""On-device batch builder.

The reference assembles every batch on the host — `GraphSampler.__getitem__` pads a dense float64 adjacency per
graph, the DataLoader stacks them, `train.py:197-201` converts to float32 and copies `[B,N,N]` to the GPU: 20 MB per
DD step, more PCIe time than the whole fused forward + backward.  Here the dataset is held as edge lists; a batch
is shipped as a few int32 arrays (edges, offsets, node labels) and `dp_build_batch` writes the padded dense batch
in device memory, in the layout the encoders take (`adj [B,N,N]`, one-hot `feats [B,N,F]`, `num_nodes`).

    ds = EdgeListDataset.from_tu_graphs(read_tu_graphs(datadir, "ENZYMES", max_nodes=100))
    builder = DeviceBatchBuilder(ds, max_nodes=100, feat_dim=3, device="cuda")
    batch = builder.build(indices)        # same keys as tu_dataset.collate / the reference's DataLoader batches
    ypred = model(batch["feats"], batch["adj"], batch["num_nodes"], assign_x=batch["assign_feats"])
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _lib


class EdgeListDataset:
    """Graphs as concatenated int32 edge lists (each undirected edge once, i < j) + node labels + graph labels."""

    def __init__(self, edge_src, edge_dst, edge_ptr, node_label, node_ptr, graph_label):
        self.edge_src = np.ascontiguousarray(edge_src, dtype=np.int32)
        self.edge_dst = np.ascontiguousarray(edge_dst, dtype=np.int32)
        self.edge_ptr = np.ascontiguousarray(edge_ptr, dtype=np.int64)
        self.node_label = np.ascontiguousarray(node_label, dtype=np.int32)
        self.node_ptr = np.ascontiguousarray(node_ptr, dtype=np.int64)
        self.graph_label = np.ascontiguousarray(graph_label, dtype=np.int64)

    def __len__(self):
        return len(self.graph_label)

    def num_nodes(self, g: int) -> int:
        return int(self.node_ptr[g + 1] - self.node_ptr[g])

    @classmethod
    def from_tu_graphs(cls, graphs) -> "EdgeListDataset":
        """From tu_dataset.TUGraph objects (dense symmetric 0/1 adjacency, optional node labels)."""
        srcs, dsts, labels, eptr, nptr, gl = [], [], [], [0], [0], []
        for g in graphs:
            i, j = np.nonzero(np.triu(g.adj, 1))
            srcs.append(i.astype(np.int32))
            dsts.append(j.astype(np.int32))
            eptr.append(eptr[-1] + len(i))
            n = g.num_nodes
            labels.append(np.asarray(g.node_label, dtype=np.int32) if g.node_label is not None
                          else np.zeros(n, dtype=np.int32))
            nptr.append(nptr[-1] + n)
            gl.append(g.label)
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dtype=dt)   # noqa: E731
        return cls(cat(srcs, np.int32), cat(dsts, np.int32), eptr, cat(labels, np.int32), nptr, gl)

    def gather(self, indices: Sequence[int]):
        """Host arrays of one batch: (src, dst, edge_ptr[B+1], node_label, node_ptr[B+1], graph_label, max_edges)."""
        idx = np.asarray(indices, dtype=np.int64)
        e0, e1 = self.edge_ptr[idx], self.edge_ptr[idx + 1]
        n0, n1 = self.node_ptr[idx], self.node_ptr[idx + 1]
        src = np.concatenate([self.edge_src[a:b] for a, b in zip(e0, e1)]) if len(idx) else np.zeros(0, np.int32)
        dst = np.concatenate([self.edge_dst[a:b] for a, b in zip(e0, e1)]) if len(idx) else np.zeros(0, np.int32)
        lab = np.concatenate([self.node_label[a:b] for a, b in zip(n0, n1)]) if len(idx) else np.zeros(0, np.int32)
        edge_ptr = np.concatenate([[0], np.cumsum(e1 - e0)]).astype(np.int32)
        node_ptr = np.concatenate([[0], np.cumsum(n1 - n0)]).astype(np.int32)
        max_edges = int((e1 - e0).max()) if len(idx) else 0
        return src, dst, edge_ptr, lab, node_ptr, self.graph_label[idx], max_edges


class DeviceBatchBuilder:
    _RING = 4          # pinned staging buffers in rotation (a buffer is reused only after its copy has completed)

    _MODES = {"default": 0, "id": 1, "deg-num": 2, "deg": 3}

    def __init__(self, dataset: EdgeListDataset, max_nodes: int, feat_dim: int, device="cuda",
                 features: str = "default", assign_feat: str = "default"):
        """`features` / `assign_feat` are the sampler's modes (graph_sampler.py:33-59, 85-87); `feat_dim` is the
        width of the one-hot node labels.  'struct' (clustering coefficients) is host-only: tu_dataset.collate."""
        if features not in self._MODES:
            raise ValueError(f"feature mode {features!r} is not built on the device (default | id | deg-num | deg); "
                             "use tu_dataset.collate for 'struct'")
        if assign_feat not in ("default", "id"):
            raise ValueError(f"unknown assign_feat {assign_feat!r} (default | id)")
        self.ds, self.N, self.F = dataset, int(max_nodes), int(feat_dim)
        self.mode, self.assign_id = self._MODES[features], assign_feat == "id"
        self.Fout = {0: self.F, 1: self.N, 2: 1, 3: 11 + self.F}[self.mode]
        self.device = torch.device(device)
        self._stage = [None] * self._RING       # (pinned int32 tensor, event)
        self._turn = 0

    def _to_device(self, parts):
        """ONE host->device copy of all int32 arrays back to back, through a reusable pinned buffer (pinning per
        call costs milliseconds — more than the whole training step)."""
        total = sum(len(p) for p in parts)
        dev = self.device
        if dev.type != "cuda":
            return torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(dev)
        slot = self._turn % self._RING
        self._turn += 1
        buf, ev = self._stage[slot] if self._stage[slot] is not None else (None, None)
        if buf is None or buf.numel() < total:
            buf = torch.empty(max(total, 1 << 16), dtype=torch.int32).pin_memory()
            ev = torch.cuda.Event()
        else:
            ev.synchronize()                     # the copy that last used this buffer is done
        view = buf.numpy()
        o = 0
        for p in parts:
            view[o:o + len(p)] = p
            o += len(p)
        packed = torch.empty(total, device=dev, dtype=torch.int32)
        packed.copy_(buf[:total], non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        self._stage[slot] = (buf, ev)
        return packed

    def build(self, indices: Sequence[int], check: bool = True, packed: bool = False) -> Dict[str, object]:
        """Device tensors `adj`, `feats` (= `assign_feats`), `label`, plus `num_nodes` as the host int array the
        reference passes (train.py:200) and `num_nodes_device`.  The batch is validated on the host (sizes, label
        range) before anything is shipped; `check` additionally reads back the kernel's skipped-entry counter.
        packed=True: `adj` is a `PackedAdjacency` (bf16 rows the kernels multiply from, written directly by
        dp_build_batch_packed; the edge lists are undirected, so A and A^T share one buffer) — no fp32 [B,N,N] batch is
        written; the encoders' forward takes it in place of the dense tensor."""
        src, dst, edge_ptr, lab, node_ptr, glabel, max_edges = self.ds.gather(indices)
        B = len(glabel)
        if B == 0:
            raise ValueError("empty batch")
        n = np.diff(node_ptr)
        if int(n.max()) > self.N:
            raise ValueError(f"a graph of the batch has {int(n.max())} nodes > max_nodes={self.N} "
                             "(the reference's loader drops such graphs, load_data.py:79)")
        if self.mode in (0, 3) and len(lab) and (int(lab.min()) < 0 or int(lab.max()) >= self.F):
            raise ValueError(f"node label outside [0, {self.F})")
        dev = self.device
        parts = [src, dst, edge_ptr, lab, node_ptr]
        offs = np.cumsum([0] + [len(p) for p in parts])
        shipped = self._to_device(parts)
        base = shipped.data_ptr()
        # (an empty array's pointer is never dereferenced)
        out = self.emit([base + 4 * int(offs[i]) for i in range(len(parts))], B, max_edges, packed)
        out["_shipped"] = shipped                      # keeps the device copy of the edge lists alive with the batch
        if check and int(out["errors"].item()) != 0:
            raise RuntimeError(f"dp_build_batch skipped {int(out['errors'].item())} out-of-range entries")
        out["label"] = torch.from_numpy(np.ascontiguousarray(glabel)).to(dev)
        out["num_nodes"] = n.astype(np.int32)
        return out

    def emit(self, ptrs, B: int, max_edges: int, packed: bool) -> Dict[str, object]:
        """The device side of `build`: allocate the batch and enqueue dp_build_batch (or _packed) on the current stream.
        ptrs = addresses of (src, dst, edge_ptr[B+1], node_label, node_ptr[B+1]) as int32 arrays the DEVICE can read —
        device memory, or pinned host memory (the kernels then fetch the lists over PCIe themselves, which is what a
        captured training step does: train_step.CapturedTrainStep)."""
        from .encoders import PackedAdjacency
        lib = _lib.load()
        dev = self.device
        feats = torch.empty(B, self.N, self.Fout, device=dev, dtype=torch.float32)
        assign = torch.empty(B, self.N, self.N + self.Fout, device=dev, dtype=torch.float32) if self.assign_id else None
        degree = torch.empty(B * self.N, device=dev, dtype=torch.int32) if self.mode >= 2 else None
        nn_dev = torch.empty(B, device=dev, dtype=torch.int32)
        errors = torch.empty(1, device=dev, dtype=torch.int32)
        _lib.require_gpu_tensor(feats, "batch")
        if packed:
            pk = torch.empty(B, self.N, lib.dp_adj_pack_ld(self.N), device=dev, dtype=torch.int16)
            _lib.check(lib.dp_build_batch_packed(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], pk.data_ptr(),
                                                 pk.data_ptr(), feats.data_ptr(), _lib.ptr(assign), nn_dev.data_ptr(),
                                                 errors.data_ptr(), _lib.ptr(degree), B, self.N, self.F, self.mode, 1,
                                                 max_edges, _lib.current_stream()), "dp_build_batch_packed")
            adj = PackedAdjacency(pk, pk, self.N)
        else:
            adj = torch.empty(B, self.N, self.N, device=dev, dtype=torch.float32)
            _lib.check(lib.dp_build_batch(ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], adj.data_ptr(), feats.data_ptr(),
                                          _lib.ptr(assign), nn_dev.data_ptr(), errors.data_ptr(), _lib.ptr(degree), B,
                                          self.N, self.F, self.mode, 1, max_edges, _lib.current_stream()),
                       "dp_build_batch")
        return {"adj": adj, "feats": feats, "assign_feats": assign if assign is not None else feats,
                "num_nodes_device": nn_dev, "errors": errors}
