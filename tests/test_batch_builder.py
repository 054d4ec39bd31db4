"""N1 — on-device batch builder against the host collate of tu_dataset (the reference's batch layout)."""
import numpy as np
import pytest
import torch

from graph_pooling_amd.batch_builder import DeviceBatchBuilder, EdgeListDataset
from graph_pooling_amd.tu_dataset import TUGraph, collate


def _random_graphs(count, n_max, n_labels, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.integers(1, n_max + 1))
        a = np.triu((rng.random((n, n)) < 0.15).astype(np.float32), 1)
        out.append(TUGraph(a + a.T, rng.integers(0, n_labels, n), int(rng.integers(0, 4))))
    return out


def test_edge_list_dataset_gather_matches_graphs():
    graphs = _random_graphs(12, 30, 5, seed=1)
    ds = EdgeListDataset.from_tu_graphs(graphs)
    assert len(ds) == 12
    idx = [7, 0, 11, 3]
    src, dst, eptr, lab, nptr, gl, max_e = ds.gather(idx)
    assert eptr[0] == 0 and nptr[0] == 0 and len(eptr) == 5
    for k, g in enumerate(idx):
        a = np.zeros_like(graphs[g].adj)
        s, d = src[eptr[k]:eptr[k + 1]], dst[eptr[k]:eptr[k + 1]]
        a[s, d] = 1
        a[d, s] = 1
        np.testing.assert_array_equal(a, graphs[g].adj)
        np.testing.assert_array_equal(lab[nptr[k]:nptr[k + 1]], graphs[g].node_label)
        assert gl[k] == graphs[g].label
    assert max_e == max(eptr[k + 1] - eptr[k] for k in range(4))


def test_builder_rejects_oversized_graphs_and_bad_labels_on_the_host():
    graphs = _random_graphs(4, 30, 5, seed=2)
    ds = EdgeListDataset.from_tu_graphs(graphs)
    big = max(g.num_nodes for g in graphs)
    with pytest.raises(ValueError):
        DeviceBatchBuilder(ds, max_nodes=big - 1, feat_dim=5, device="cpu").build(range(4))
    with pytest.raises(ValueError):
        DeviceBatchBuilder(ds, max_nodes=big, feat_dim=2, device="cpu").build(range(4))


@pytest.mark.gpu
@pytest.mark.parametrize("count,n_max,F_,N", [(20, 100, 3, 100), (7, 37, 6, 64), (3, 1, 2, 8), (20, 500, 89, 500)])
def test_device_batch_is_bit_identical_to_host_collate(count, n_max, F_, N):
    graphs = _random_graphs(count, n_max, F_, seed=count + N)
    ds = EdgeListDataset.from_tu_graphs(graphs)
    idx = list(reversed(range(count)))
    ref = collate([graphs[i] for i in idx], N, F_)
    got = DeviceBatchBuilder(ds, N, F_, "cuda").build(idx)
    assert torch.equal(got["adj"].cpu(), torch.from_numpy(ref["adj"]))
    assert torch.equal(got["feats"].cpu(), torch.from_numpy(ref["feats"]))
    np.testing.assert_array_equal(got["num_nodes"], ref["num_nodes"])
    np.testing.assert_array_equal(got["num_nodes_device"].cpu().numpy(), ref["num_nodes"])
    np.testing.assert_array_equal(got["label"].cpu().numpy(), ref["label"])
    assert got["assign_feats"] is got["feats"]


@pytest.mark.gpu
@pytest.mark.parametrize("features,assign_feat", [("id", "default"), ("deg-num", "default"), ("deg", "default"),
                                                  ("default", "id"), ("deg", "id")])
def test_device_feature_modes_match_host_collate(features, assign_feat):
    count, n_max, F_, N = 9, 40, 4, 48
    graphs = _random_graphs(count, n_max, F_, seed=21)
    hub = np.zeros((20, 20), dtype=np.float32)                 # a degree above the one-hot cap of 10
    hub[0, 1:] = hub[1:, 0] = 1
    graphs.append(TUGraph(hub, np.zeros(20, dtype=np.int64), 1))
    ds = EdgeListDataset.from_tu_graphs(graphs)
    ref = collate(graphs, N, F_, features=features, assign_feat=assign_feat)
    got = DeviceBatchBuilder(ds, N, F_, "cuda", features=features, assign_feat=assign_feat).build(range(len(graphs)))
    assert torch.equal(got["adj"].cpu(), torch.from_numpy(ref["adj"]))
    assert torch.equal(got["feats"].cpu(), torch.from_numpy(ref["feats"]))
    assert torch.equal(got["assign_feats"].cpu(), torch.from_numpy(ref["assign_feats"]))
    np.testing.assert_array_equal(got["num_nodes"], ref["num_nodes"])


def test_struct_mode_is_host_only():
    ds = EdgeListDataset.from_tu_graphs(_random_graphs(2, 8, 3, seed=1))
    with pytest.raises(ValueError):
        DeviceBatchBuilder(ds, 8, 3, "cpu", features="struct")


@pytest.mark.gpu
def test_device_built_batch_drives_the_encoder_like_the_host_batch():
    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
    graphs = _random_graphs(6, 40, 4, seed=9)
    ds = EdgeListDataset.from_tu_graphs(graphs)
    ref = collate(graphs, 40, 4)
    got = DeviceBatchBuilder(ds, 40, 4, "cuda").build(range(6))
    torch.manual_seed(0)
    model = SoftPoolingGcnEncoder(40, 4, 8, 8, 4, 3, 8, assign_ratio=0.25, linkpred=False).cuda()
    y_dev = model(got["feats"], got["adj"], got["num_nodes"], assign_x=got["assign_feats"])
    y_host = model(torch.from_numpy(ref["feats"]).cuda(), torch.from_numpy(ref["adj"]).cuda(), ref["num_nodes"],
                   assign_x=torch.from_numpy(ref["feats"]).cuda())
    assert torch.equal(y_dev, y_host)
