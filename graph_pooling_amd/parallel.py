"""Data parallelism over graphs: one process per GPU, one RCCL all-reduce per step.

Every contraction of the DiffPool path is per graph, so a batch shards over ranks with exactly one
exchange per step: the sum of the flat fp32 gradient buffer (42 KB ENZYMES / 75 KB DD / <= 1 MB ER —
SURVEY.md §8(e)).  That is latency-bound, far below the xGMI link rate, so the whole model is ONE
bucket and ONE collective (`torch.distributed` backend "nccl" = RCCL on ROCm; "gloo" on CPU for tests);
there is nothing to overlap it with.  The reference has no distributed code at all (train.py:624 picks
a single device).

BatchNorm: apply_bn normalises per node index over the whole batch
(encoders.py:1048-1052).  Two modes:
  * local (default, throughput): statistics over the rank's shard — a sharded step equals "world_size
    independent reference steps with averaged gradients", not one step on the concatenated batch;
  * sync_bn=True (parity): every BatchNorm site all-gathers its per-row partials (mean, M2 forward;
    sum dx, sum dx*xhat backward: [B, n, G, 2] floats, 80 KB at the DD shape) before combining them, and the
    link loss is normalised by the global sum of n_b^2 — the averaged gradients are then those of ONE
    reference step on the concatenated batch.  Costs 2 (L-1) (1 + 2P) extra small collectives per step.

What has run where (tests/test_gpu_parallel.py): parity of both modes against the oracle is pinned with TWO ranks over
GLOO sharing the box's one GPU (RCCL refuses two ranks per device), at a small, a packed and the DD shard shape.  The
RCCL branches (ReduceOp.AVG, all_gather_into_tensor, a collective captured into the step's hipGraph) have run with ONE
rank only (`force=True` below keeps every collective on at world size 1); no multi-rank RCCL run exists yet.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class DataParallelEncoder:
    """Wraps one of the encoders of graph_pooling_amd.encoders.

        dp = DataParallelEncoder(model)        # after model.cuda(); broadcasts rank 0's parameters
        loss.backward(); dp.reduce_gradients(); clip_grad_norm_(...); optimizer.step()
    """

    def __init__(self, model, process_group: Optional[dist.ProcessGroup] = None, broadcast: bool = True,
                 sync_bn: bool = False, force: bool = False):
        """force=True keeps every collective path on at world size 1 (the parameter broadcast, the gradient
        all-reduce, the sync-BN all-gathers and the link-normaliser all-reduce all execute, over one rank): the way to
        run the RCCL code on a one-GPU box."""
        self.model = model
        self.group = process_group
        self.force = bool(force) and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # How the mean over ranks is formed is decided ONCE, here, identically on every rank: RCCL averages inside
        # the all-reduce (ncclAvg), gloo (CPU tests, one-GPU rehearsals) has no AVG -> SUM, then one scale.  A failing
        # collective is never caught and retried: it propagates and the process exits non-zero.
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.reduce_op = dist.ReduceOp.AVG if self.backend == "nccl" else dist.ReduceOp.SUM
        if dist.is_initialized() and dist.get_rank(process_group) == 0:
            print(f"[graph_pooling_amd.parallel] world {self.world}, backend {self.backend}: gradients averaged with "
                  f"{'AVG inside the all-reduce' if self.backend == 'nccl' else 'SUM + scale'}; "
                  f"BatchNorm statistics {'synchronised' if sync_bn else 'local to a rank'}", flush=True)
        device = next(model.parameters()).device
        model._ensure_flat(device)
        if sync_bn and (self.world > 1 or self.force):
            model._sync_bn = SyncBatchNormExchange(process_group, self.world, force=self.force)
            model._plans = {}                       # plans built before carry no exchange callback
        if broadcast and (self.world > 1 or self.force):
            self.sync_parameters()

    def __getattr__(self, name):
        return getattr(self.model, name)

    def __call__(self, *a, **k):
        return self.model(*a, **k)

    def sync_parameters(self, src: int = 0):
        """Every rank starts from rank `src`'s flat parameter buffer (one broadcast)."""
        m = self.model
        m._ensure_flat(next(m.parameters()).device)
        dist.broadcast(m._flat, src=src, group=self.group)

    def _average(self, flat):
        """Mean over ranks in place: ONE collective (mode fixed in __init__; errors propagate)."""
        dist.all_reduce(flat, op=self.reduce_op, group=self.group)
        if self.reduce_op != dist.ReduceOp.AVG:
            flat.div_(self.world)

    def _aliased_flat_grad(self):
        m = self.model
        g = getattr(m, "_last_flat_grad", None)
        if g is None:
            return None
        base = g.data_ptr()
        for p, (off, numel, _) in zip(m._flat_params, m._flat_index):
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return None
        return g

    def reduce_gradients(self):
        """Average gradients over ranks with ONE all-reduce of the flat buffer."""
        if self.world == 1 and not self.force:
            return
        m = self.model
        flat = self._aliased_flat_grad()
        if flat is not None:
            # the autograd engine kept our views: p.grad already aliases the flat buffer
            self._average(flat)
            return
        device = m._flat.device
        flat = torch.zeros(m._flat.numel(), device=device, dtype=torch.float32)
        for p, (off, numel, _) in zip(m._flat_params, m._flat_index):
            if p.grad is not None:
                flat[off:off + numel].copy_(p.grad.reshape(-1))
        self._average(flat)
        for p, (off, numel, shape) in zip(m._flat_params, m._flat_index):
            if p.grad is None:
                p.grad = flat[off:off + numel].view(shape)
            else:
                p.grad.copy_(flat[off:off + numel].view(shape))


class SyncBatchNormExchange:
    """The collectives of sync-BN mode, called from inside dp_encoder_forward / backward through the plan's exchange
    callback (encoders._Plan): an all-gather of a BatchNorm site's row partials on the current stream."""

    def __init__(self, group, world, force=False):
        self.group, self.world = group, world
        self.active = world > 1 or force          # force: the collectives run over a single rank too
        self.error = None
        self.calls = 0

    def fail(self, exc):
        """A collective raised inside the library's callback: the peers are (or will be) blocked in the same
        collective, so this rank must not carry on as if it could recover.  Record the error, abort the process group
        so the peers' collectives fail instead of hanging, and let the caller raise.  Never retried."""
        self.error = exc
        try:
            abort = getattr(dist.distributed_c10d, "_abort_process_group", None)
            if abort is not None:
                abort(self.group)
        except Exception:          # noqa: BLE001 — the abort is best effort; the original error is what gets reported
            pass

    def all_gather(self, dst, src):
        # rank-major = batch-major: rank r's block lands at rows [r * B, (r + 1) * B) of the gathered partials
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(dst, src, group=self.group)
        else:
            dist.all_gather(list(dst.chunk(self.world)), src, group=self.group)
        self.calls += 1

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def collective_capture_works(device, world: int, rank: int) -> bool:
    """Capture and replay one all-reduce on a throw-away RCCL communicator; True iff it ran and gave the right sum on
    every rank.  A failed capture can leave a communicator unusable, so the attempt is made on a group of its own: the
    real one survives to run the collective eagerly.  Every rank takes the same answer (MIN over ranks)."""
    import sys
    ok = 0.0
    try:
        grp = dist.new_group(ranks=list(range(world)), backend="nccl")
        t = torch.full((1024,), float(rank + 1), device=device)
        dist.all_reduce(t, group=grp)                       # communicator set-up happens eagerly
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            t.fill_(float(rank + 1))
            dist.all_reduce(t, group=grp)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        t.fill_(float(rank + 1))
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            dist.all_reduce(t, group=grp)
        t.fill_(float(rank + 1))
        g.replay()
        torch.cuda.synchronize()
        ok = 1.0 if abs(float(t[0]) - world * (world + 1) / 2) < 1e-3 else 0.0
    except Exception as e:                                   # noqa: BLE001
        print(f"[graph_pooling_amd.parallel] rank {rank}: a captured all-reduce is not available here "
              f"({type(e).__name__}: {e})", file=sys.stderr)
        ok = 0.0
        try:
            torch.cuda.synchronize()
        except Exception:                                    # noqa: BLE001
            pass
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0.5)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """Contiguous, equal shards of a collated batch dict (adj / feats / num_nodes / label / assign_feats)."""
    B = len(batch["num_nodes"])
    if B % world != 0:
        raise ValueError(f"batch size {B} is not divisible by world size {world}")
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    return {k: v[sl] for k, v in batch.items()}
