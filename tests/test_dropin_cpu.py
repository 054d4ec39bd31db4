"""`dropin.install()` is what lets the reference's train.py run unchanged (`import encoders`, train.py:493-508): the
four module names must resolve to the HIP-backed modules, and the constructor calls train.py makes — with its argparse
namespace passed as `args=` (only `.bias` is read, encoders.py:986-988) — must build.  CPU only: construction needs no
GPU; a forward on CPU tensors must fail loudly (there is no fallback)."""
import argparse
import sys

import pytest
import torch


@pytest.fixture()
def installed():
    saved = {k: sys.modules.get(k) for k in ("encoders", "set2set", "aggregators", "graphsage")}
    import graph_pooling_amd.dropin as dropin
    enc = dropin.install()
    yield enc
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def _args():
    # the fields benchmark_task_val reads for the constructors (train.py:520-611 defaults)
    return argparse.Namespace(hidden_dim=20, output_dim=20, num_classes=6, num_gc_layers=3, assign_ratio=0.1, num_pool=1,
                              bn=True, dropout=0.0, linkpred=False, bias=True, method="soft-assign")


def test_install_aliases_the_reference_module_names(installed):
    import graph_pooling_amd.aggregators as A
    import graph_pooling_amd.encoders as E
    import graph_pooling_amd.graphsage as G
    import graph_pooling_amd.set2set as S
    import encoders, set2set, aggregators, graphsage          # noqa: E401 — what train.py / encoders.py import
    assert encoders is E and set2set is S and aggregators is A and graphsage is G
    assert installed is E
    for name in ("SoftPoolingGcnEncoder", "GcnSet2SetEncoder", "GcnEncoderGraph", "GraphConv"):
        assert hasattr(encoders, name)
    assert hasattr(set2set, "Set2Set") and hasattr(aggregators, "MeanAggregator")
    assert hasattr(graphsage, "SupervisedGraphSage")


@pytest.mark.parametrize("bias", [True, False])
def test_train_py_constructor_calls_build(installed, bias):
    import encoders
    args = _args()
    args.bias = bias
    max_num_nodes, input_dim, assign_input_dim = 100, 3, 3
    # train.py:493-498
    m1 = encoders.SoftPoolingGcnEncoder(
        max_num_nodes, input_dim, args.hidden_dim, args.output_dim, args.num_classes, args.num_gc_layers,
        args.hidden_dim, assign_ratio=args.assign_ratio, num_pooling=args.num_pool, bn=args.bn, dropout=args.dropout,
        linkpred=args.linkpred, args=args, assign_input_dim=assign_input_dim)
    # train.py:501-503
    m2 = encoders.GcnSet2SetEncoder(input_dim, args.hidden_dim, args.output_dim, args.num_classes, args.num_gc_layers,
                                    bn=args.bn, dropout=args.dropout, args=args)
    # train.py:506-508
    m3 = encoders.GcnEncoderGraph(input_dim, args.hidden_dim, args.output_dim, args.num_classes, args.num_gc_layers,
                                  bn=args.bn, dropout=args.dropout, args=args)
    # train.py:345-357 (no args object)
    m4 = encoders.SoftPoolingGcnEncoder(max_num_nodes, input_dim, args.hidden_dim, args.output_dim, 2, args.num_gc_layers,
                                        args.hidden_dim, assign_ratio=args.assign_ratio, num_pooling=args.num_pool,
                                        bn=args.bn, linkpred=args.linkpred, assign_input_dim=assign_input_dim)
    for m in (m1, m2, m3):
        has_bias = any(k.startswith("conv_first.bias") for k in m.state_dict())
        assert has_bias == bias
        assert len(list(m.parameters())) > 0           # train.py:173 hands these to Adam
    assert "conv_first.bias" in m4.state_dict()
    assert m1.assign_dims == [10] and m1.label_dim == 6
    # what train.py calls next needs a GPU: on CPU tensors the modules refuse instead of falling back
    x, adj = torch.zeros(2, 100, 3), torch.zeros(2, 100, 100)
    with pytest.raises(RuntimeError, match="GPU"):
        m1(x, adj, [5, 7], assign_x=x)
