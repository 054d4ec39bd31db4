"""Shared helpers of the GPU parity tests.

Tolerances: forward outputs rtol 1e-4 / atol 1e-5; parameter gradients rtol 1e-3 with an absolute
floor of 2e-5 x the tensor's largest entry (fp32 reduction-order noise measured against an fp64 run is 1e-6..1e-5 of
the largest entry, on the HIP path and in the torch-CPU oracle alike — tools/grad_anchor_probe.py).

Discrete decisions: torch.max routes the max-readout gradient to the winning row.  Rows of a pooled level tie to ~1e-8
(soft assignments are nearly uniform at init), so the winner of such a tie can differ between two fp32 evaluation
orders; each flip moves whole gradient entries by ~1e-3 relative although both results are valid gradients.  Tests
against the oracle therefore run the oracle with the winners the HIP forward recorded (`gpu_winners`), after the
forward comparison has shown that those winners hold the maximum up to rounding (the readout features agree)."""
import torch


def close(a, b, rtol=1e-4, atol=1e-5):
    a = a.detach().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a)
    b = b.detach().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b)
    assert torch.isfinite(a).all() and torch.isfinite(b).all(), 'non-finite values in a parity check'
    torch.testing.assert_close(a.float(), b.float(), rtol=rtol, atol=atol)


def grads_close(model, ref_grads, rtol=1e-3, atol_rel=2e-5):
    named = dict(model.named_parameters())
    assert set(named) == set(ref_grads), set(named) ^ set(ref_grads)
    for k, p in named.items():
        assert p.grad is not None, k
        g = ref_grads[k]
        scale = float(g.abs().max())
        try:
            close(p.grad, g, rtol=rtol, atol=max(1e-7, atol_rel * scale))
        except AssertionError as e:
            raise AssertionError(f"gradient of {k} (largest reference entry {scale:.3e}): {e}") from None


def gpu_winners(model, levels):
    """Rows the HIP forward's max readout picked, per level: int32 [B, D] (-1: a masked zero row holds the maximum)."""
    return [model.saved_activation(j, "readout_argmax").clone().cpu() for j in range(levels)]
