"""Host-side logic that needs no GPU: the encoder config the modules hand to the C ABI (flat-parameter offsets,
per-level widths, dropout mask slots), validated by the library's own dry-run sizing, and the argument checks of
the optimiser / batch-builder entry points (they return before any launch)."""
import ctypes as C

import numpy as np
import pytest
import torch

from graph_pooling_amd import _lib
from graph_pooling_amd.encoders import GcnEncoderGraph, GcnSet2SetEncoder, SoftPoolingGcnEncoder


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def _cfg(model, B, N):
    model._ensure_flat(torch.device("cpu"))
    cfg = model._build_cfg(B, N)
    segs, total = model._fill_dropout(cfg, B)
    return cfg, segs, total


def test_softpool_cfg_matches_the_flat_parameter_buffer(lib):
    N, F_, H, Cc, B = 100, 3, 20, 6, 20
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.1, linkpred=True)
    cfg, segs, total = _cfg(model, B, N)
    assert (cfg.B, cfg.N, cfg.num_pooling) == (B, N, 1)
    assert list(cfg.n_nodes[:2]) == [100, 10]                       # K = int(100 * 0.1), encoders.py:1203
    assert list(cfg.embed[0].dims[:4]) == [F_, H, H, H]
    assert list(cfg.assign[0].dims[:4]) == [F_, H, H, 10]
    assert list(cfg.embed[1].dims[:4]) == [3 * H, H, H, H]          # after-pool GCN eats the concat (D = 60)
    assert cfg.pred_dims[0] == 2 * 3 * H and cfg.pred_dims[cfg.n_pred] == Cc
    assert cfg.n_params == sum(p.numel() for p in model.parameters()) == model._flat.numel()
    # every offset addresses the tensor it names inside the flat buffer
    offs = model._offsets()
    flat = model._flat
    for p, (off, numel, shape) in zip(model._flat_params, model._flat_index):
        assert offs[id(p)] == off and p.data_ptr() == flat.data_ptr() + 4 * off and tuple(p.shape) == shape
    assert cfg.embed[0].w_off[0] == offs[id(model.conv_first.weight)]
    assert cfg.assign_pred_w_off[0] == offs[id(model.assign_pred.weight)]
    assert total == 0 and segs == []                                 # dropout 0: no mask slots
    assert all(cfg.embed[j].drop_off[l] == -1 for j in range(2) for l in range(_lib.DP_MAX_LAYERS))
    # the library accepts the config and sizes its buffers from it (dry run, no GPU)
    assert lib.dp_sizeof_encoder_cfg() == C.sizeof(_lib.EncoderCfg)
    assert lib.dp_encoder_save_bytes(C.byref(cfg)) > 0
    assert lib.dp_encoder_workspace_bytes(C.byref(cfg)) > 0


def test_dropout_slots_follow_the_reference_wiring(lib):
    """`dropout` reaches only the conv_block layers of the after-pool stacks (encoders.py:1180-1183, 1013-1016)."""
    N, F_, H, Cc, B = 64, 5, 8, 3, 4
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 4, H, assign_ratio=0.25, num_pooling=2, dropout=0.4)
    cfg, segs, total = _cfg(model, B, N)
    n1, n2 = int(cfg.n_nodes[1]), int(cfg.n_nodes[2])
    assert (n1, n2) == (16, 4)
    assert [p for _, _, p in segs] == [0.4] * 4
    assert [numel for _, numel, _ in segs] == [B * n1 * H, B * n1 * H, B * n2 * H, B * n2 * H]
    assert total == sum(numel for _, numel, _ in segs)
    for j in (1, 2):
        assert cfg.embed[j].drop_off[0] == -1 and cfg.embed[j].drop_off[3] == -1      # conv_first / conv_last
        assert cfg.embed[j].drop_off[1] >= 0 and cfg.embed[j].drop_off[2] > cfg.embed[j].drop_off[1]
    assert all(cfg.embed[0].drop_off[l] == -1 for l in range(4))                     # level 0: never (D9)
    assert all(cfg.assign[j].drop_off[l] == -1 for j in range(2) for l in range(4))  # assign stacks: never
    ws_with = lib.dp_encoder_workspace_bytes(C.byref(cfg))
    plain = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 4, H, assign_ratio=0.25, num_pooling=2, dropout=0.0)
    cfg0, _, _ = _cfg(plain, B, N)
    assert ws_with > lib.dp_encoder_workspace_bytes(C.byref(cfg0))                    # masked-input scratch


def test_base_and_set2set_cfgs(lib):
    base = GcnEncoderGraph(7, 12, 10, 4, 3, pred_hidden_dims=[9], concat=True, bn=True)
    cfg, _, _ = _cfg(base, 5, 30)
    assert cfg.num_pooling == 0 and cfg.readout == 0 and cfg.n_pred == 2
    assert list(cfg.embed[0].dims[:4]) == [7, 12, 12, 10] and cfg.pred_dims[0] == 12 * 2 + 10
    assert lib.dp_encoder_workspace_bytes(C.byref(cfg)) > 0
    s2s = GcnSet2SetEncoder(3, 20, 20, 6, 3)
    cfg2, _, _ = _cfg(s2s, 20, 100)
    assert cfg2.readout == 1 and all(cfg2.s2s_off[i] >= 0 for i in range(6))
    assert lib.dp_encoder_save_bytes(C.byref(cfg2)) > lib.dp_encoder_save_bytes(C.byref(cfg))


def test_invalid_cfgs_are_rejected_before_any_launch(lib):
    model = SoftPoolingGcnEncoder(32, 3, 8, 8, 2, 3, 8, assign_ratio=0.25)
    cfg, _, _ = _cfg(model, 4, 32)
    cfg.embed[1].drop_off[0] = 0                       # a mask on conv_first: not a reference configuration
    assert lib.dp_encoder_workspace_bytes(C.byref(cfg)) == 0
    assert b"dropout" in lib.dp_last_error_string()
    cfg, _, _ = _cfg(model, 4, 32)
    cfg.n_nodes[1] = 0
    assert lib.dp_encoder_save_bytes(C.byref(cfg)) == 0


def test_optimiser_and_batch_builder_argument_checks(lib):
    z = (C.c_float * 4)()
    ws = (C.c_char * 4096)()
    p = C.addressof(z)
    assert lib.dp_clip_adam_step(p, p, p, p, 4, 0, 1e-3, 0.9, 0.999, 1e-8, 2.0, None, C.addressof(ws), 4096, None) < 0
    assert lib.dp_clip_adam_step(None, p, p, p, 4, 1, 1e-3, 0.9, 0.999, 1e-8, 2.0, None, C.addressof(ws), 4096, None) < 0
    assert lib.dp_clip_adam_step(p, p, p, p, 4, 1, 1e-3, 0.9, 1.5, 1e-8, 2.0, None, C.addressof(ws), 4096, None) < 0
    assert lib.dp_clip_adam_workspace_bytes() >= 1024
    i4 = (C.c_int * 4)()
    q = C.addressof(i4)
    assert lib.dp_build_batch(q, q, q, None, q, p, p, None, q, q, None, 2, 8, 3, 0, 1, 4, None) < 0   # feats, no labels
    assert lib.dp_build_batch(q, q, q, q, q, None, p, None, q, q, None, 2, 8, 3, 0, 1, 4, None) < 0   # adj is NULL
    assert lib.dp_build_batch(q, q, q, q, q, p, p, None, q, q, None, 2, 8, 3, 3, 1, 4, None) < 0      # deg mode, no ws
    assert lib.dp_build_batch(q, q, q, q, q, p, p, None, q, q, q, 2, 8, 3, 7, 1, 4, None) < 0         # unknown mode


def test_supervised_graphsage_head_on_a_stub_encoder():
    """graphsage.py:7-26 with the D10 repairs: scores = enc(nodes) @ weight, cross-entropy on the raw scores."""
    from graph_pooling_amd.graphsage import SupervisedGraphSage

    class Enc(torch.nn.Module):
        embed_dim = 5

        def __init__(self):
            super().__init__()
            self.table = torch.nn.Parameter(torch.randn(11, 5))

        def forward(self, nodes):
            return self.table[torch.as_tensor(nodes)]

    torch.manual_seed(0)
    head = SupervisedGraphSage(3, Enc())
    assert tuple(head.weight.shape) == (5, 3) and set(dict(head.named_parameters())) == {"weight", "enc.table"}
    with pytest.raises(RuntimeError, match="GPU"):          # no CPU path (numerics: tests/test_gpu_ops.py)
        head([0, 4, 7, 10])
    with pytest.raises(ValueError):
        SupervisedGraphSage(0, Enc())


def test_captured_train_step_checks_its_arguments_before_touching_the_gpu():
    """train_step.CapturedTrainStep needs the device-counted optimizer and has no link loss (the packed adjacency has no
    fp32 form for it): both are refused up front, on any host."""
    import numpy as np
    from graph_pooling_amd.batch_builder import DeviceBatchBuilder, EdgeListDataset
    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
    from graph_pooling_amd.optim import FusedClipAdam
    from graph_pooling_amd.train_step import CapturedTrainStep
    from graph_pooling_amd.tu_dataset import TUGraph
    a = np.zeros((3, 3), dtype=np.float32)
    a[0, 1] = a[1, 0] = 1
    ds = EdgeListDataset.from_tu_graphs([TUGraph(a, np.zeros(3, dtype=np.int64), 0)] * 4)
    builder = DeviceBatchBuilder(ds, 64, 2, "cpu")
    model = SoftPoolingGcnEncoder(64, 2, 4, 4, 2, 3, 4, assign_ratio=0.25, linkpred=False)
    with pytest.raises(ValueError, match="device_step_counter"):
        CapturedTrainStep(model, FusedClipAdam(model), builder, 2)
    link = SoftPoolingGcnEncoder(64, 2, 4, 4, 2, 3, 4, assign_ratio=0.25, linkpred=True)
    with pytest.raises(ValueError, match="link-prediction"):
        CapturedTrainStep(link, FusedClipAdam(link, device_step_counter=True), builder, 2)
