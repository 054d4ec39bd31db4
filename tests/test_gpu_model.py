"""GPU parity of the nn.Module surface (graph_pooling_amd.encoders) against (a) the golden vectors the
reference's own classes produced (tests/golden, oracle/make_golden.py) and (b) the CPU oracle on
seeded synthetic batches up to BASELINE.json's DD shape.  The modules call libdiffpool_hip.so through
the C ABI (dp_encoder_forward/backward, dp_loss_forward/backward).

Tolerances (fp32 kernels, reduction-order differences only) and the handling of max-readout ties:
tests/parity.py.
"""
import numpy as np
import pytest
import torch

from graph_pooling_amd.encoders import GcnEncoderGraph, GcnSet2SetEncoder, SoftPoolingGcnEncoder
from graph_pooling_amd.set2set import Set2Set
from oracle import diffpool_oracle as O
from tests.parity import close, grads_close, gpu_winners

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.mark.parametrize("name", ["g4_softpool_n16_f3", "g5_softpool_n16_f3_link", "g4_softpool_n100_f89",
                                  "g5_softpool_n100_f89_link", "g9_enzymes_batch"])
def test_softpool_against_reference_golden(name, golden):
    a, params, grads = golden(name)
    x = T(a["x"])
    B, N, F_ = x.shape
    if "adj" in a:
        adj = T(a["adj"])
    else:
        adj = T(np.unpackbits(a["adj_bits"], axis=-1)[..., :N].astype(np.float32))
    linkpred = "link_loss" in a
    H = params["conv_first.weight"].shape[1]
    E = params["conv_last.weight"].shape[1]
    K = params["assign_pred.weight"].shape[0]
    Cc = params["pred_model.2.weight"].shape[0]
    model = SoftPoolingGcnEncoder(N, F_, H, E, Cc, 3, H, assign_ratio=K / N + 1e-9, linkpred=linkpred)
    assert model.assign_dims == [K]
    model.load_state_dict(params)            # same keys and shapes as the reference (Appendix D)
    model = model.cuda()
    xd, ad = x.cuda(), adj.cuda()
    ypred = model(xd, ad, a["num_nodes"], assign_x=xd)
    win = gpu_winners(model, 2)
    close(ypred, a["ypred"])
    close(model.assign_tensor, a["assign"], 1e-4, 1e-6)
    label = T(a["label"]).cuda()
    loss = model.loss(ypred, label, ad, a["num_nodes"]) if linkpred else model.loss(ypred, label)
    close(loss, a["loss"], 1e-5, 1e-6)
    if linkpred:
        close(model.link_loss, a["link_loss"], 1e-5, 1e-6)
    loss.backward()
    # Strict: every gradient tensor within rtol 1e-3 (floor 2e-5 x its largest entry) of the REFERENCE's own fp32 numbers.
    # The only tensors excused from that are listed, per fixture, in ANCHORED_GRADS: ill-conditioned entries (a bias
    # gradient in front of a BatchNorm is a sum that nearly cancels — torch-CPU on another host already moves G4's
    # conv_block2.0.bias by 3e-3 relative) — and those must instead be no further from an fp64 run of the oracle (HIP
    # winners forced) than 4x the reference's own fp32 gradient is.  Any OTHER tensor that misses the strict tolerance
    # fails the test, so a regression cannot hide behind the anchored criterion.
    strict_fail = {}
    named = dict(model.named_parameters())
    assert set(named) == set(grads)
    for k, p in named.items():
        try:
            grads_close(_One(k, p), {k: grads[k]})
        except AssertionError as e:
            strict_fail[k] = str(e)
    allowed = ANCHORED_GRADS.get(name, set())
    extra = set(strict_fail) - allowed
    assert not extra, f"{name}: tensors outside the allow-list miss rtol 1e-3: " + "; ".join(strict_fail[k] for k in extra)
    if strict_fail:
        print(f"[anchored] {name}: {sorted(strict_fail)} checked against the fp64 oracle instead of rtol 1e-3")
        Pm = {k: v.clone().double().requires_grad_(True) for k, v in params.items()}
        yo, inter = O.softpool_forward(Pm, x.double(), adj.double(), a["num_nodes"], x.double(), winners=win)
        lo, _ = O.softpool_loss(yo, T(a["label"]), inter["assign_0"], adj.double(), a["num_nodes"], linkpred)
        lo.backward()
        close(yo, a["ypred"])                              # the forced winners hold the maximum: same forward
        for k in strict_fail:
            g64 = Pm[k].grad
            e_gpu = float((named[k].grad.detach().cpu().double() - g64).abs().max())
            e_ref = float((grads[k].double() - g64).abs().max())
            assert e_gpu <= 4 * e_ref + 3e-7 * float(g64.abs().max()), \
                f"{k}: |hip - fp64| {e_gpu:.3e} against the reference's own |fp32 - fp64| {e_ref:.3e}"


# gradient tensors of the golden fixtures that may be checked against the fp64 oracle instead of rtol 1e-3 (see above)
ANCHORED_GRADS = {
    "g4_softpool_n16_f3": {"conv_block2.0.bias"},
}


class _One:
    """grads_close() on a single named parameter."""

    def __init__(self, k, p):
        self._kp = (k, p)

    def named_parameters(self):
        return [self._kp]


@pytest.mark.parametrize("tag", ["concat", "addself", "nobn"])
def test_base_encoder_against_reference_golden(tag, golden):
    a, params, grads = golden(f"g6_base_{tag}")
    concat, bn, nh = [int(v) for v in a["cfg"]]
    x, adj = T(a["x"]), T(a["adj"])
    B, N, F_ = x.shape
    H = params["conv_first.weight"].shape[1]
    E = params["conv_last.weight"].shape[1]
    hidden = [params["pred_model.0.weight"].shape[0]] if nh else []
    Cc = params["pred_model.2.weight" if nh else "pred_model.weight"].shape[0]
    model = GcnEncoderGraph(F_, H, E, Cc, 3, pred_hidden_dims=hidden, concat=bool(concat), bn=bool(bn))
    model.load_state_dict(params)
    model = model.cuda()
    ypred = model(x.cuda(), adj.cuda(), a["num_nodes"])
    close(ypred, a["ypred"])
    loss = model.loss(ypred, T(a["label"]).cuda())
    close(loss, a["loss"], 1e-5, 1e-6)
    loss.backward()
    grads_close(model, grads)


def _oracle_run(params, x, adj, nn_, label, linkpred, num_pooling=1, winners=None):
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(P, x, adj, nn_, x, num_pooling=num_pooling, winners=winners)
    lo, link = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, linkpred)
    lo.backward()
    return yo, inter, lo, {k: v.grad for k, v in P.items()}


@pytest.mark.parametrize("B,N,F_,H,Cc,ratio,p,linkpred,tag", [
    (20, 100, 3, 20, 6, 0.1, 0.10, True, "S-ENZ"),       # BASELINE configs[0] shape
    (20, 500, 89, 20, 2, 0.1, 0.02, False, "S-DD"),      # BASELINE configs[1] shape (the metric's workload)
    (20, 500, 89, 20, 2, 0.1, 0.02, True, "S-DD+link"),
    (5, 67, 11, 12, 3, 0.25, 0.15, True, "odd"),
    (3, 256, 16, 20, 3, 0.6, 0.05, True, "wide-K"),       # K = 154 clusters: operands wider than 128 -> wide bf16 kernel
    (132, 512, 8, 12, 2, 0.05, 0.02, False, "big-B"),     # 528 row tiles of 128: wide kernel with the fused tail
])
def test_softpool_against_oracle_synthetic(B, N, F_, H, Cc, ratio, p, linkpred, tag):
    n_min = max(1, N // 10)
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=n_min, p=p, seed=1, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, linkpred=linkpred)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=0, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    xd, ad = x.cuda(), adj.cuda()
    ypred = model(xd, ad, nn_, assign_x=xd)
    win = gpu_winners(model, 2)
    loss = model.loss(ypred, label.cuda(), ad, nn_) if linkpred else model.loss(ypred, label.cuda())
    loss.backward()
    yfree, _ = O.softpool_forward(params, x, adj, nn_, x)
    close(ypred, yfree)                                   # forward against the oracle's own arg-max
    yo, inter, lo, go = _oracle_run(params, x, adj, nn_, label, linkpred, winners=win)
    close(ypred, yo)
    close(model.assign_tensor, inter["assign_0"], 1e-4, 1e-6)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, go)


def test_unmasked_batch_and_full_graphs():
    # batch_num_nodes = None (no masking anywhere) and n_b == N for every graph
    B, N, F_, H, Cc = 3, 24, 4, 8, 2
    x, adj, _, label = O.make_batch(B, N, F_, n_min=N, p=0.2, seed=3, n_classes=Cc, onehot=False)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, linkpred=True)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=4, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    ypred = model(x.cuda(), adj.cuda(), None, assign_x=x.cuda())
    win = gpu_winners(model, 2)
    loss = model.loss(ypred, label.cuda(), adj.cuda(), None)
    loss.backward()
    close(ypred, O.softpool_forward(params, x, adj, None, x)[0])
    yo, inter, lo, go = _oracle_run(params, x, adj, None, label, True, winners=win)
    close(ypred, yo)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, go)


def test_multi_pool_against_oracle_PARITY_UNPINNED():
    """num_pooling = 2: the reference cannot execute this, so this is pinned
    only against the build's own CPU restatement of the chosen semantics."""
    B, N, F_, H, Cc = 4, 64, 6, 10, 3
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=8, p=0.1, seed=5, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, num_pooling=2, linkpred=True)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    mine = O.softpool_param_shapes(max_num_nodes=N, input_dim=F_, hidden_dim=H, embedding_dim=H, label_dim=Cc,
                                   num_layers=3, assign_hidden_dim=H, assign_ratio=0.25, num_pooling=2)
    assert mine == shapes
    params = O.init_params(shapes, seed=6, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    ypred = model(x.cuda(), adj.cuda(), nn_, assign_x=x.cuda())
    win = gpu_winners(model, 3)
    loss = model.loss(ypred, label.cuda(), adj.cuda(), nn_)
    loss.backward()
    close(ypred, O.softpool_forward(params, x, adj, nn_, x, num_pooling=2)[0])
    yo, inter, lo, go = _oracle_run(params, x, adj, nn_, label, True, num_pooling=2, winners=win)
    close(ypred, yo)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, go)


def _multi_level_case(B, N, F_, H, Cc, ratio, P, linkpred, *, p_edge, n_min, onehot, seed):
    """One num_pooling = P model against the oracle: ypred, loss, every level's S / X' / A' and all gradients."""
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=n_min, p=p_edge, seed=seed, n_classes=Cc, onehot=onehot)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, num_pooling=P, linkpred=linkpred)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert shapes == O.softpool_param_shapes(max_num_nodes=N, input_dim=F_, hidden_dim=H, embedding_dim=H,
                                             label_dim=Cc, num_layers=3, assign_hidden_dim=H, assign_ratio=ratio,
                                             num_pooling=P)
    params = O.init_params(shapes, seed=seed + 1, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    xd, ad = x.cuda(), adj.cuda()
    ypred = model(xd, ad, nn_, assign_x=xd)
    loss = model.loss(ypred, label.cuda(), ad, nn_) if linkpred else model.loss(ypred, label.cuda())
    lv = [{w: model.saved_activation(j, w).clone() for w in ("assign", "xpool", "adjpool")} for j in range(P)]
    win = gpu_winners(model, P + 1)
    loss.backward()

    close(ypred, O.softpool_forward(params, x, adj, nn_, x, num_pooling=P)[0])
    Pm = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    yo, inter = O.softpool_forward(Pm, x, adj, nn_, x, num_pooling=P, want_intermediates=True, winners=win)
    lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj, nn_, linkpred)
    lo.backward()
    for j in range(P):
        close(lv[j]["assign"], inter[f"assign_{j}"], 1e-4, 1e-6)
        close(lv[j]["xpool"], inter[f"xpool_{j}"], 1e-4, 1e-5)
        sc = float(inter[f"adjpool_{j}"].abs().max())
        close(lv[j]["adjpool"], inter[f"adjpool_{j}"], 1e-4, 1e-5 * max(1.0, sc))
    close(ypred, yo)
    close(model.assign_tensor, inter["assign_0"], 1e-4, 1e-6)
    close(loss, lo, 1e-4, 1e-6)
    grads_close(model, {k: v.grad for k, v in Pm.items()})
    return model


def test_er_two_level_pooling_against_oracle_PARITY_UNPINNED():
    """BASELINE configs[2] (S-ER: N=1024, F=64 N(0,1) features, n_b = N, p = 0.01, two pooling levels K = 256 -> 64,
    H=E=20) at B = 4, where the CPU oracle takes seconds.  Level 1 runs with n = 256 nodes on a dense, non-bf16-exact
    A' (the plan's fp32 paths), its assign stack is fed X', and both levels' X'/A' are ticketed split-K products.
    The reference cannot execute num_pooling > 1 (encoders.py:1273, :1214): pinned against
    the build's CPU restatement only."""
    _multi_level_case(4, 1024, 64, 20, 2, 0.25, 2, True, p_edge=0.01, n_min=1024, onehot=False, seed=41)


@pytest.mark.parametrize("linkpred", [False, True])
def test_three_level_pooling_enzymes_shape_against_oracle_PARITY_UNPINNED(linkpred):
    """BASELINE configs[4], second half: SoftPoolingGcnEncoder(num_pooling=3) on the S-ENZ batch (B=20, N=100, F=3,
    ratio 0.25 -> K = 25, 6, 1).  Reference cannot run it (Appendix B): pinned against the oracle only."""
    m = _multi_level_case(20, 100, 3, 20, 6, 0.25, 3, linkpred, p_edge=0.10, n_min=10, onehot=True, seed=43)
    assert m.assign_dims == [25, 6, 1]


def test_er_full_size_properties():
    """S-ER at BASELINE's full size (B=256, N=1024, F=64, K = 256 -> 64): size-independent properties where the CPU
    oracle would take minutes — everything finite; every row of each level's S is a distribution; the pooled
    adjacency conserves mass (rows of S sum to 1  =>  sum A'_j = sum A_j); the forward is bit-reproducible."""
    B, N, F_, H, Cc = 256, 1024, 64, 20, 2
    g = torch.Generator(device="cuda").manual_seed(3)
    up = torch.triu((torch.rand(B, N, N, device="cuda", generator=g) < 0.01).float(), diagonal=1)
    adj = up + up.transpose(1, 2)
    del up
    x = torch.randn(B, N, F_, device="cuda", generator=g)
    label = torch.randint(0, Cc, (B,), device="cuda", generator=g)
    nn_ = torch.full((B,), N, dtype=torch.int32, device="cuda")
    torch.manual_seed(0)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, num_pooling=2, linkpred=False).cuda()
    runs = []
    for it in range(3):
        model.zero_grad(set_to_none=True)
        y = model(x, adj, nn_, assign_x=x)
        acts = [{w: model.saved_activation(j, w).clone() for w in ("assign", "adjpool")} for j in range(2)]
        runs.append((y.clone(), acts))
        if it == 0:
            loss = model.loss(y, label)
            loss.backward()
            assert torch.isfinite(loss)
            for k, p in model.named_parameters():
                assert p.grad is not None and torch.isfinite(p.grad).all(), k
                assert float(p.grad.abs().max()) > 0.0, k
    y0, a0 = runs[0]
    assert torch.isfinite(y0).all()
    mass = adj.sum(dim=(1, 2))
    for j in range(2):
        S = a0[j]["assign"]
        assert torch.isfinite(S).all() and float(S.min()) >= 0.0
        close(S.sum(dim=2), torch.ones(S.shape[:2]), 1e-5, 1e-5)
        Ap = a0[j]["adjpool"]
        assert torch.isfinite(Ap).all()
        close(Ap.sum(dim=(1, 2)), mass, 2e-4, 1e-2)
    for y, acts in runs[1:]:
        assert torch.equal(y, y0)
        for j in range(2):
            assert torch.equal(acts[j]["assign"], a0[j]["assign"]) and torch.equal(acts[j]["adjpool"], a0[j]["adjpool"])


@pytest.mark.parametrize("seed", [1, 2, 5, 7])
@pytest.mark.parametrize("tag,B,N,F_,H,Cc,ratio,p,linkpred", [
    ("S-DD", 20, 500, 89, 20, 2, 0.1, 0.02, False),
    ("S-ENZ+link", 20, 100, 3, 20, 6, 0.1, 0.10, True),
])
def test_gradients_no_worse_than_fp32_oracle_vs_fp64(tag, B, N, F_, H, Cc, ratio, p, linkpred, seed):
    """fp64-anchored gradient check (instead of a loose relative tolerance): the fp32 torch-CPU oracle is itself only
    an approximation of the exact gradient, so the HIP path is held to the oracle's own distance from an fp64 run of the
    same restatement:  max|g_gpu - g64| <= 4 * max|g_oracle32 - g64| + 3e-7 * max|g64|  for EVERY parameter tensor
    (the floor is 5 ulp of the tensor's largest entry: for a bias gradient that is a 20-term sum the oracle's own error
    can by luck be under one ulp, and 4x that is not a meaningful bound).

    One thing has to be equal on both sides first: the discrete decisions.  torch.max routes the readout gradient to
    the winning row, and at a pooled level rows tie to ~1e-8 (soft assignments are nearly uniform at init), so the
    winner of such a tie differs between fp32 and fp64 runs — in the oracle too (tools/grad_anchor_probe.py: 0-5 such
    flips per batch, each moving whole gradient entries by ~1e-3 relative, on the HIP path and the fp32 oracle alike).
    So the winners the HIP forward recorded are (1) checked to be legitimate — the value at the recorded row is
    within rounding of the fp64 maximum — and (2) forced into the fp32 and fp64 oracle runs."""
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=max(1, N // 10), p=p, seed=seed, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=ratio, linkpred=linkpred)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed - 1, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    xd, ad = x.cuda(), adj.cuda()
    ypred = model(xd, ad, nn_, assign_x=xd)
    winners = gpu_winners(model, 2)
    loss = model.loss(ypred, label.cuda(), ad, nn_) if linkpred else model.loss(ypred, label.cuda())
    loss.backward()

    def oracle(dtype, win):
        Pm = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
        yo, inter = O.softpool_forward(Pm, x.to(dtype), adj.to(dtype), nn_, x.to(dtype), winners=win)
        lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj.to(dtype), nn_, linkpred)
        lo.backward()
        return float(lo), inter["readout"].detach().double(), {k: v.grad.double() for k, v in Pm.items()}
    _, feat_free, _ = oracle(torch.float64, None)           # fp64 with its own arg-max
    l32, _, g32 = oracle(torch.float32, winners)
    l64, feat_forced, g64 = oracle(torch.float64, winners)
    # (1) every recorded winner holds the maximum up to rounding
    assert float((feat_forced - feat_free).abs().max()) <= 2e-6 * max(1.0, float(feat_free.abs().max()))
    assert abs(float(loss) - l64) <= 4 * abs(l32 - l64) + 1e-6 * abs(l64)
    for k, pm in model.named_parameters():
        gg = pm.grad.detach().cpu().double()
        assert torch.isfinite(gg).all(), k
        e_gpu = float((gg - g64[k]).abs().max())
        e_o32 = float((g32[k] - g64[k]).abs().max())
        bound = 4 * e_o32 + 3e-7 * float(g64[k].abs().max())
        assert e_gpu <= bound, f"{tag} seed {seed} {k}: |gpu-fp64| {e_gpu:.3e} > 4*|oracle32-fp64| {e_o32:.3e} + 3e-7*scale"


def test_adam_two_steps_golden(golden):
    """train.py:173,209-210 — Adam(lr 1e-3) + clip_grad_norm(2.0) on top of the HIP fwd/bwd."""
    a, params, _ = golden("g10_adam_two_steps")
    x, adj = T(a["x"]).cuda(), T(a["adj"]).cuda()
    label = T(a["label"]).cuda()
    B, N, F_ = x.shape
    model = SoftPoolingGcnEncoder(N, F_, 8, 8, 6, 3, 8, assign_ratio=0.25, linkpred=True)
    model.load_state_dict(params)
    model = model.cuda()
    opt = torch.optim.Adam(model.parameters(), lr=0.001)
    for step in range(2):
        model.zero_grad()
        ypred = model(x, adj, a["num_nodes"], assign_x=x)
        loss = model.loss(ypred, label, adj, a["num_nodes"])
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 2.0)
        opt.step()
        close(loss, a["losses"][step], 1e-5, 1e-6)
    sd = model.state_dict()
    for k, v in a["after"].items():
        close(sd[k], v, 1e-4, 1e-6)


def test_eval_mode_and_no_grad_match_train_forward():
    # apply_bn always uses batch statistics, .eval() changes nothing
    B, N, F_, H, Cc = 4, 32, 5, 8, 3
    x, adj, nn_, _ = O.make_batch(B, N, F_, n_min=3, p=0.2, seed=7, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, linkpred=False).cuda()
    y1 = model(x.cuda(), adj.cuda(), nn_, assign_x=x.cuda())
    model.eval()
    with torch.no_grad():
        y2 = model(x.cuda(), adj.cuda(), nn_, assign_x=x.cuda())
    close(y1, y2, 0, 0)
    labels = model.predict(x.cuda(), adj.cuda(), nn_, assign_x=x.cuda())
    assert labels.dtype == torch.int64 and labels.is_cuda and labels.shape == (B,)
    assert torch.equal(labels, y1.argmax(dim=1))


def test_predict_matches_reference_argmax_on_the_enzymes_batch(golden):
    """N3 / train.py:30-58: evaluate() = forward + torch.max(ypred, 1).  predict() runs a DP_MODE_EVAL forward whose
    head launch writes the class ids; on the real-ENZYMES fixture (G9) they must equal the arg-max of the reference's
    own ypred, and the logits of the eval forward must be the training forward's bit for bit."""
    a, params, _ = golden("g9_enzymes_batch")
    x = T(a["x"])
    B, N, F_ = x.shape
    adj = T(np.unpackbits(a["adj_bits"], axis=-1)[..., :N].astype(np.float32))
    model = SoftPoolingGcnEncoder(N, F_, 20, 20, 6, 3, 20, assign_ratio=0.1, linkpred=True)
    model.load_state_dict(params)
    model = model.cuda().eval()
    xd, ad = x.cuda(), adj.cuda()
    labels = model.predict(xd, ad, a["num_nodes"], assign_x=xd)
    assert labels.dtype == torch.int64 and labels.shape == (B,)
    assert torch.equal(labels.cpu(), T(a["ypred"]).argmax(dim=1))
    y_train = model(xd, ad, a["num_nodes"], assign_x=xd)              # grad mode: DP_MODE_TRAIN forward
    with torch.no_grad():
        y_eval = model(xd, ad, a["num_nodes"], assign_x=xd)
    assert torch.equal(y_train, y_eval) and torch.equal(labels, y_eval.argmax(dim=1))
    # a plan whose prediction head is not the fused kernel (Set2Set readout) takes the arg-max kernel instead
    s2s = GcnSet2SetEncoder(F_, 8, 8, 6, 3).cuda()
    assert torch.equal(s2s.predict(xd, ad, a["num_nodes"]), s2s(xd, ad, a["num_nodes"]).argmax(dim=1))


def test_two_forwards_before_their_backwards():
    """Each training forward keeps its own activations, and only the latest forward may hand its pre-cleared
    accumulators to a backward pass (DP_MODE_TRAIN / prezeroed): fwd(a), fwd(b), bwd(a), bwd(b) — and a second backward
    through a retained graph — must give the gradients of separate steps."""
    B, N, F_, H, Cc = 5, 160, 6, 10, 3
    xa, adja, nna, la = O.make_batch(B, N, F_, n_min=20, p=0.05, seed=11, n_classes=Cc)
    xb, adjb, nnb, lb = O.make_batch(B, N, F_, n_min=20, p=0.05, seed=12, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.2, linkpred=True).cuda()

    def grads_of(loss, **kw):
        model.zero_grad(set_to_none=True)
        loss.backward(**kw)
        return {k: p.grad.clone() for k, p in model.named_parameters()}

    def run(x, adj, nn_, label):
        xd, ad = x.cuda(), adj.cuda()
        y = model(xd, ad, nn_, assign_x=xd)
        return model.loss(y, label.cuda(), ad, nn_)
    ga = grads_of(run(xa, adja, nna, la))
    gb = grads_of(run(xb, adjb, nnb, lb))
    loss_a = run(xa, adja, nna, la)
    loss_b = run(xb, adjb, nnb, lb)
    ga2 = grads_of(loss_a, retain_graph=True)
    gb2 = grads_of(loss_b)
    ga3 = grads_of(loss_a)                         # second backward through the retained graph
    for k in ga:                                   # every tensor bit for bit (round 3: no float atomics on this path)
        close(ga2[k], ga[k], 0, 0)
        close(gb2[k], gb[k], 0, 0)
        close(ga3[k], ga[k], 0, 0)


def test_linearity_of_pooling_at_full_size():
    """Size-independent property at the DD shape: with S fixed, X' = S^T Z is linear in Z and
    A' = S^T A S is linear in A (checked through dp_pool_fwd on the full B=20, N=500 batch)."""
    from graph_pooling_amd import _lib
    lib = _lib.load()
    B, n, K, D = 20, 500, 50, 60
    g = torch.Generator().manual_seed(0)
    S_ = torch.softmax(torch.randn(B, n, K, generator=g), -1).cuda()
    Z1, Z2 = torch.randn(B, n, D, generator=g).cuda(), torch.randn(B, n, D, generator=g).cuda()
    A1 = (torch.rand(B, n, n, generator=g) < 0.02).float().cuda()
    A2 = (torch.rand(B, n, n, generator=g) < 0.02).float().cuda()
    st = torch.cuda.current_stream().cuda_stream

    def pool(Z, A):
        X = torch.empty(B, K, D, device="cuda")
        Ap = torch.empty(B, K, K, device="cuda")
        T_ = torch.empty(B, K, n, device="cuda")
        _lib.check(lib.dp_pool_fwd(S_.data_ptr(), Z.data_ptr(), D, A.data_ptr(), X.data_ptr(), Ap.data_ptr(),
                                   T_.data_ptr(), B, n, K, D, st))
        return X, Ap
    X1, P1 = pool(Z1, A1)
    X2, P2 = pool(Z2, A2)
    X3, P3 = pool(Z1 + 2 * Z2, A1 + 2 * A2)
    close(X3, X1 + 2 * X2, 1e-4, 1e-4)
    close(P3, P1 + 2 * P2, 1e-4, 1e-4)
    # every row of S sums to 1  =>  sum(A') == sum(A)  (mass conservation of S^T A S)
    close(P1.sum(dim=(1, 2)), A1.sum(dim=(1, 2)), 1e-4, 1e-2)


# ------------------------------------------------------------------ A10 Set2Set
@pytest.mark.parametrize("n", [7, 100])
def test_set2set_against_reference_golden(n, golden):
    """Set2Set.forward / backward (set2set.py:32-57) through dp_set2set_fwd / dp_set2set_bwd.
    The recurrence runs n steps, so reduction-order differences compound: rtol 2e-3 on gradients."""
    a, params, grads = golden(f"g7_set2set_n{n}")
    d = a["emb"].shape[2]
    m = Set2Set(d, 2 * d)
    m.load_state_dict(params)
    m = m.cuda()
    emb = T(a["emb"]).cuda().requires_grad_(True)
    out = m(emb)
    close(out, a["out"], 1e-4, 1e-5)
    (out * T(a["gout"]).cuda()).sum().backward()
    close(emb.grad, a["gemb"], 2e-3, 2e-5)
    named = dict(m.named_parameters())
    for k, g in grads.items():
        close(named[k].grad, g, 2e-3, max(2e-5, 2e-4 * float(g.abs().max())))


def test_set2set_encoder_against_reference_golden(golden):
    a, params, grads = golden("g7_set2set_encoder")
    x, adj = T(a["x"]), T(a["adj"])
    B, N, F_ = x.shape
    H = params["conv_first.weight"].shape[1]
    E = params["conv_last.weight"].shape[1]
    Cc = params["pred_model.weight"].shape[0]
    model = GcnSet2SetEncoder(F_, H, E, Cc, 3)
    model.load_state_dict(params)
    model = model.cuda()
    ypred = model(x.cuda(), adj.cuda(), a["num_nodes"])
    close(ypred, a["ypred"])
    loss = model.loss(ypred, T(a["label"]).cuda())
    close(loss, a["loss"], 1e-5, 1e-6)
    loss.backward()
    grads_close(model, grads)


def test_set2set_enzymes_shape_against_oracle():
    # S-S2S: B=20, N=100, D=60 -> LSTM(120 -> 60), 100 sequential steps
    B, n, d = 20, 100, 60
    g = torch.Generator().manual_seed(0)
    emb = torch.randn(B, n, d, generator=g) * 0.3
    emb[:, 70:] = 0.0
    m = Set2Set(d, 2 * d)
    params = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    e_d = emb.cuda().requires_grad_(True)
    out = m(e_d)
    gout = torch.randn(B, d, generator=g)
    (out * gout.cuda()).sum().backward()
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    e_o = emb.clone().requires_grad_(True)
    oo = O.set2set_forward(e_o, P)
    (oo * gout).sum().backward()
    close(out, oo, 1e-4, 1e-5)
    close(e_d.grad, e_o.grad, 2e-3, 2e-5)
    named = dict(m.named_parameters())
    for k, v in P.items():
        close(named[k].grad, v.grad, 2e-3, max(2e-5, 2e-4 * float(v.grad.abs().max())))


def test_step_is_capturable_in_a_hip_graph_and_replays_bit_identically():
    """No call of the path allocates, frees or synchronises: forward + loss + backward captured
    once in a hipGraph must replay to the same bits as the eager step, also after the inputs change in place."""
    B, N, F_, H, Cc = 6, 160, 7, 12, 3
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=20, p=0.05, seed=5, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.2, linkpred=True).cuda()
    xd, ad, ld = x.cuda(), adj.cuda(), label.cuda()
    nd = torch.from_numpy(nn_).cuda()                   # device num_nodes: no H2D inside the captured region

    def step():
        model.zero_grad(set_to_none=True)
        y = model(xd, ad, nd, assign_x=xd)
        loss = model.loss(y, ld, ad, nd)
        loss.backward()
        return y, loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    model.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        y_static, loss_static = step()
    grads_static = {k: p.grad for k, p in model.named_parameters()}

    for seed in (5, 9):                                  # same inputs, then new inputs written in place
        x2, adj2, nn2, label2 = O.make_batch(B, N, F_, n_min=20, p=0.05, seed=seed, n_classes=Cc)
        xd.copy_(x2); ad.copy_(adj2); ld.copy_(label2); nd.copy_(torch.from_numpy(nn2))
        g.replay()
        torch.cuda.synchronize()
        y_g, loss_g = y_static.clone(), loss_static.clone()
        grads_g = {k: v.clone() for k, v in grads_static.items()}
        y_e, loss_e = step()                             # eager, same inputs
        close(y_g, y_e, 0, 0)
        close(loss_g, loss_e, 0, 0)
        for k, p in model.named_parameters():
            # every gradient tensor bit for bit: the level-0 kernels combine per-block partials in block order, the
            # pooled-level kernels their per-wave bias partials in wave order (round 3; float atomics before)
            close(grads_g[k], p.grad, 0, 0)


@pytest.mark.parametrize("linkpred", [False, True])
def test_loss_backward_fast_path_equals_every_other_way_of_calling_it(linkpred):
    """`model.loss(...).backward()` takes a fast path (no seed fill, no cross-entropy backward launch: encoders._Loss).
    The same gradients must come out of every other way a caller can start the backward pass — a derived root,
    an explicit `gradient=`, `torch.autograd.grad` — and an explicit gradient of 2 must double them."""
    B, N, F_, H, Cc = 5, 40, 4, 8, 3
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=5, p=0.15, seed=11, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, linkpred=linkpred)
    params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=2, bias_scale=0.1)
    model.load_state_dict(params)
    model = model.cuda()
    xd, ad, ld = x.cuda(), adj.cuda(), label.cuda()

    def loss_of():
        model.zero_grad(set_to_none=True)
        y = model(xd, ad, nn_, assign_x=xd)
        return model.loss(y, ld, ad, nn_) if linkpred else model.loss(y, ld)

    def grads_after(run):
        loss = loss_of()
        out = run(loss)
        torch.cuda.synchronize()
        return out if out is not None else {k: p.grad.clone() for k, p in model.named_parameters()}

    fast = grads_after(lambda l: l.backward())
    assert type(loss_of()).__name__ == "_Loss"
    names = [k for k, _ in model.named_parameters()]
    ways = {
        "(loss * 1).backward()": lambda l: (l * 1).backward(),
        "loss.backward(gradient=ones)": lambda l: l.backward(gradient=torch.ones((), device="cuda")),
        "autograd.grad": lambda l: dict(zip(names, torch.autograd.grad(l, list(model.parameters())))),
        "loss.sum().backward()": lambda l: l.sum().backward(),
    }
    for what, run in ways.items():
        g = grads_after(run)
        for k in names:
            try:
                close(g[k], fast[k], 1e-5, 1e-8)
            except AssertionError as e:
                raise AssertionError(f"{what}: {k}: {e}") from None
    twice = grads_after(lambda l: l.backward(gradient=torch.full((), 2.0, device="cuda")))
    for k in names:
        close(twice[k], 2 * fast[k], 1e-4, 1e-6 * max(1.0, float(fast[k].abs().max())))
    # and the oracle agrees with the fast path
    win = gpu_winners(model, 2)
    _, _, _, go = _oracle_run(params, x, adj, nn_, label, linkpred, winners=win)
    loss_of().backward()
    grads_close(model, go)


def test_gradients_against_the_oracle_with_its_own_winners():
    """The other gradient tests against the oracle force the HIP forward's max-readout winners into the oracle run
    (tests/parity.py says why).  This one does not: the oracle picks its own arg-max, on a batch chosen — by the oracle
    alone, before the GPU is asked anything — to have no near-ties: the first seed of a fixed list for which the
    oracle's unforced gradients agree between fp32 and fp64 (a flipped winner moves whole entries by ~1e-3; agreement to
    3e-5 of each tensor's largest entry means no decision depends on rounding)."""
    B, N, F_, H, Cc = 6, 40, 5, 8, 3
    chosen = None
    for seed in range(1, 13):
        x, adj, nn_, label = O.make_batch(B, N, F_, n_min=8, p=0.2, seed=seed, n_classes=Cc, onehot=False)
        model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, linkpred=True)
        params = O.init_params({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed, bias_scale=0.3)
        grads = {}
        for dt in (torch.float32, torch.float64):
            P = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
            yo, inter = O.softpool_forward(P, x.to(dt), adj.to(dt), nn_, x.to(dt))
            lo, _ = O.softpool_loss(yo, label, inter["assign_0"], adj.to(dt), nn_, True)
            lo.backward()
            grads[dt] = {k: v.grad.double() for k, v in P.items()}
        stable = all(float((grads[torch.float32][k] - grads[torch.float64][k]).abs().max())
                     <= 3e-5 * float(grads[torch.float64][k].abs().max()) + 1e-9 for k in params)
        if stable:
            chosen = (seed, x, adj, nn_, label, model, params, grads[torch.float32])
            break
    assert chosen is not None, "no seed of the list gives a batch without near-tied readout rows"
    seed, x, adj, nn_, label, model, params, g32 = chosen
    model.load_state_dict(params)
    model = model.cuda()
    xd, ad = x.cuda(), adj.cuda()
    ypred = model(xd, ad, nn_, assign_x=xd)
    loss = model.loss(ypred, label.cuda(), ad, nn_)
    loss.backward()
    close(ypred, O.softpool_forward(params, x, adj, nn_, x)[0])
    grads_close(model, {k: v.float() for k, v in g32.items()})
