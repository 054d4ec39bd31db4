"""Child process of test_grid_barrier_give_up_is_reported_not_silent (tests/test_gpu_edge_cases.py): with the TEST-ONLY
knob DP_TEST_BARRIER_FAIL=1 the whole-level kernels wait for one arrival more than will ever come, with a spin limit of
64 — the give-up path of dp_small.hip's grid barrier.  What must happen (diffpool_hip.h "Device-side failures"):

  * the step's outputs are NaN, never plausible numbers (the kernel poisons its BatchNorm statistics, but max readouts
    and ReLUs swallow NaN — the prediction-head launch therefore reads the barrier's error word and writes NaN logits);
  * the device's error word carries DP_DEVERR_BARRIER once the stream has drained, and the NEXT model-level call
    returns DP_ERR_DEVICE -> RuntimeError naming the grid barrier, before launching anything;
  * the fused optimizer refuses the NaN gradients: parameters bit-identical to before the step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("DP_TEST_BARRIER_FAIL") == "1"
from graph_pooling_amd import _lib                                    # noqa: E402
from graph_pooling_amd.encoders import SoftPoolingGcnEncoder          # noqa: E402
from graph_pooling_amd.optim import FusedClipAdam                     # noqa: E402
from oracle import diffpool_oracle as O                               # noqa: E402


def main():
    lib = _lib.load()
    B, N, F_, H, Cc = 6, 64, 5, 8, 3
    x, adj, nn_, label = O.make_batch(B, N, F_, n_min=8, p=0.1, seed=3, n_classes=Cc)
    model = SoftPoolingGcnEncoder(N, F_, H, H, Cc, 3, H, assign_ratio=0.25, linkpred=False).cuda()
    opt = FusedClipAdam(model, lr=1e-3, clip=2.0)
    xd, ad, ld = x.cuda(), adj.cuda(), label.cuda()
    assert lib.dp_device_error(0) == 0
    model._ensure_flat(xd.device)
    before = model._flat.detach().clone()
    ypred = model(xd, ad, nn_, assign_x=xd)            # the pooled level (n = 16) runs k_small_level_fwd
    torch.cuda.synchronize()
    assert not torch.isfinite(ypred).all(), ("a barrier that gave up must poison the outputs; error word "
                                             f"{lib.dp_device_error(0)}, last message: {lib.dp_last_error_string()!r}")
    assert lib.dp_device_error(0) & _lib.DEVERR_BARRIER, "the give-up did not reach the device error word"
    print("outputs poisoned, error word set", flush=True)
    try:
        model.loss(ypred, ld)
    except RuntimeError as e:
        assert "grid barrier gave up" in str(e) and f"code {_lib.ERR_DEVICE}" in str(e), str(e)
        print("next entry raised:", str(e)[:120], flush=True)
    else:
        raise AssertionError("the next model-level entry did not report the device error")
    assert lib.dp_device_error(0) == 0                 # reported once, then cleared
    # a caller that never looked: loss + backward + optimizer on the poisoned step
    ypred = model(xd, ad, nn_, assign_x=xd)
    torch.cuda.synchronize()
    lib.dp_device_error(1)                             # swallow the forward's report, as a careless caller might
    loss = model.loss(ypred, ld)
    loss.backward()
    torch.cuda.synchronize()
    lib.dp_device_error(1)
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(model._flat, before), "the optimizer applied non-finite gradients"
    assert lib.dp_device_error(0) & _lib.DEVERR_NONFINITE_GRAD
    print("optimizer refused the non-finite gradients; parameters untouched", flush=True)
    print("barrier give-up path complete", flush=True)


if __name__ == "__main__":
    main()
