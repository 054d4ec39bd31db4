"""N2 — fused clip + Adam (dp_clip_adam_step) against torch.optim.Adam + clip_grad_norm_ and the golden trajectory."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def lib():
    from graph_pooling_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("n,clip", [(1, 2.0), (1000, 0.5), (18672, 2.0), (300001, 0.0), (70000, 1e9)])
def test_flat_step_matches_torch_adam(lib, n, clip):
    from graph_pooling_amd import _lib
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=1e-3, foreach=False, fused=False)
    pd = p0.clone().cuda()
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    tn = torch.zeros(1, device="cuda")
    ws = torch.empty(lib.dp_clip_adam_workspace_bytes(), device="cuda", dtype=torch.uint8)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 if step == 2 else 0.1)
        pt.grad = grad.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([pt], clip) if clip > 0 else grad.norm()
        opt.step()
        gd = grad.clone().cuda()
        _lib.check(lib.dp_clip_adam_step(pd.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, step, 1e-3, 0.9,
                                         0.999, 1e-8, clip, tn.data_ptr(), ws.data_ptr(), ws.numel(),
                                         torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.isfinite(pd).all()
        np.testing.assert_allclose(tn.cpu().numpy()[0], float(ref_norm), rtol=2e-6)
        np.testing.assert_allclose(gd.cpu().numpy(), pt.grad.numpy(), rtol=2e-6, atol=1e-12)     # clipped in place
        np.testing.assert_allclose(pd.cpu().numpy(), pt.detach().numpy(), rtol=2e-6, atol=2e-7)


def test_argument_errors(lib):
    z = torch.zeros(4, device="cuda")
    ws = torch.empty(lib.dp_clip_adam_workspace_bytes(), device="cuda", dtype=torch.uint8)
    args = (z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 4)
    assert lib.dp_clip_adam_step(*args, 0, 1e-3, 0.9, 0.999, 1e-8, 2.0, None, ws.data_ptr(), ws.numel(), None) < 0
    assert lib.dp_clip_adam_step(*args, 1, 1e-3, 1.0, 0.999, 1e-8, 2.0, None, ws.data_ptr(), ws.numel(), None) < 0
    assert lib.dp_clip_adam_step(*args, 1, 1e-3, 0.9, 0.999, 1e-8, 2.0, None, ws.data_ptr(), 8, None) != 0


def test_fused_optimizer_follows_the_golden_adam_trajectory(golden):
    """train.py:173,209-210 reproduced with FusedClipAdam instead of torch.optim.Adam + clip_grad_norm_."""
    from graph_pooling_amd.encoders import SoftPoolingGcnEncoder
    from graph_pooling_amd.optim import FusedClipAdam
    a, params, _ = golden("g10_adam_two_steps")
    x, adj = T(a["x"]).cuda(), T(a["adj"]).cuda()
    label = T(a["label"]).cuda()
    B, N, F_ = x.shape
    model = SoftPoolingGcnEncoder(N, F_, 8, 8, 6, 3, 8, assign_ratio=0.25, linkpred=True)
    model.load_state_dict(params)
    model = model.cuda()
    opt = FusedClipAdam(model, lr=0.001, clip=2.0)
    for step in range(2):
        opt.zero_grad()
        ypred = model(x, adj, a["num_nodes"], assign_x=x)
        loss = model.loss(ypred, label, adj, a["num_nodes"])
        loss.backward()
        opt.step()
        np.testing.assert_allclose(float(loss.detach()), float(a["losses"][step]), rtol=1e-5, atol=1e-6)
    sd = model.state_dict()
    for k, v in a["after"].items():
        got = sd[k].detach().cpu().numpy()
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, np.asarray(v), rtol=1e-4, atol=1e-6)
