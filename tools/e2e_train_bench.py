"""End-to-end training-step time on a DD-shaped synthetic dataset: batch assembly + H2D + forward + loss + backward +
clip + Adam, the way the reference's train.py loop does it (host collate of the dense batch, `.cuda()`,
clip_grad_norm_, torch.optim.Adam: train.py:197-210) against the on-device batch builder + fused clip/Adam
(SURVEY.md §8(f) N1, N2).  Model math is the same HIP path in both arms.
  PYTHONPATH=. python tools/e2e_train_bench.py [--graphs 200] [--steps 100]"""
import argparse, json, time
import numpy as np
import torch

from graph_pooling_amd.batch_builder import DeviceBatchBuilder, EdgeListDataset
from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
from graph_pooling_amd.optim import FusedClipAdam
from graph_pooling_amd.tu_dataset import TUGraph, collate


def dataset(count, n_min, n_max, n_labels, p, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.integers(n_min, n_max + 1))
        a = np.triu((rng.random((n, n)) < p).astype(np.float32), 1)
        out.append(TUGraph(a + a.T, rng.integers(0, n_labels, n), int(rng.integers(0, 2))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=200)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    B, N, F_, H, Cc = 20, 500, 89, 20, 2
    graphs = dataset(args.graphs, 30, N, F_, 0.02, seed=1)
    batches = [list(range(i, i + B)) for i in range(0, args.graphs - B + 1, B)]
    dev = torch.device("cuda")

    def run(arm):
        torch.manual_seed(0)
        model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.1, linkpred=False).cuda()
        if arm == "reference-style":
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        else:
            opt = FusedClipAdam(model, lr=1e-3, clip=2.0)
            builder = DeviceBatchBuilder(EdgeListDataset.from_tu_graphs(graphs), N, F_, dev)

        def step(idx):
            if arm == "reference-style":
                b = collate([graphs[i] for i in idx], N, F_)                       # GraphSampler + DataLoader collate
                adj = torch.from_numpy(b["adj"]).float().cuda()                    # train.py:197-201
                x = torch.from_numpy(b["feats"]).float().cuda()
                label = torch.from_numpy(b["label"]).long().cuda()
                nn_ = b["num_nodes"]
            else:
                b = builder.build(idx, check=False)
                adj, x, label, nn_ = b["adj"], b["feats"], b["label"], b["num_nodes_device"]
            model.zero_grad(set_to_none=True)
            ypred = model(x, adj, nn_, assign_x=x)
            loss = model.loss(ypred, label)
            loss.backward()
            if arm == "reference-style":
                torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
            opt.step()

        for i in range(10):
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    def run_captured():
        """build -> forward -> loss -> backward -> clip + Adam as ONE hipGraph launch per step (train_step.py)."""
        from graph_pooling_amd.train_step import CapturedTrainStep
        torch.manual_seed(0)
        model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.1, linkpred=False).cuda()
        opt = FusedClipAdam(model, lr=1e-3, clip=2.0, device_step_counter=True)
        builder = DeviceBatchBuilder(EdgeListDataset.from_tu_graphs(graphs), N, F_, dev)
        step = CapturedTrainStep(model, opt, builder, B)
        for i in range(10):
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(batches[i % len(batches)])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        assert step.skipped_entries() == 0
        return dt

    out = {arm: round(run(arm), 3) for arm in ("reference-style", "device-builder+fused-optimizer")}
    out["captured: packed builder + model + fused optimizer in one hipGraph"] = round(run_captured(), 3)
    out["unit"] = "ms per training step (B=20, N_max=500, F=89), eager, host loop included"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
