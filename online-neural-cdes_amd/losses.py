"""NaN-masked per-time-step losses for ``NeuralCDE(return_sequences=True)`` (online prediction: a NaN label marks a time
step after the series has ended; the reference trains such tasks through its loss wrapper,
/root/reference/experiments/ingredients/metrics.py:26-58).

Stream-ordered formulation: the reference gathers the valid positions with a boolean index, which on a GPU costs a
``nonzero`` + host synchronisation per step; here the invalid positions are zero-weighted inside one elementwise pass and
the mean is taken over the device-side count, so nothing waits on the host between the fused solve kernels.
"""
import torch
import torch.nn.functional as F

_POINTWISE = {
    "mse": lambda p, y: (p - y) ** 2,
    "l1": lambda p, y: (p - y).abs(),
    "bce_logits": lambda p, y: F.binary_cross_entropy_with_logits(p, y, reduction="none"),
}


def masked_mean(preds, labels, kind="mse"):
    """Mean of the pointwise loss over the positions where ``labels`` is not NaN (0 if there are none)."""
    if kind not in _POINTWISE:
        raise ValueError("kind must be one of %s" % sorted(_POINTWISE))
    valid = labels == labels                          # NaN != NaN
    y = torch.where(valid, labels, torch.zeros((), dtype=labels.dtype, device=labels.device))
    per = _POINTWISE[kind](preds, y) * valid.to(preds.dtype)
    return per.sum() / valid.sum().clamp_min(1).to(preds.dtype)


class MaskedTemporalLoss(torch.nn.Module):
    """``kind`` in {mse, rmse, l1, bce_logits}; rmse = sqrt(masked mse + eps)."""

    def __init__(self, kind="mse", eps=1e-6):
        super().__init__()
        if kind != "rmse" and kind not in _POINTWISE:
            raise ValueError("kind must be one of %s" % (sorted(_POINTWISE) + ["rmse"]))
        self.kind, self.eps = kind, eps

    def forward(self, preds, labels):
        if self.kind == "rmse":
            return torch.sqrt(masked_mean(preds, labels, "mse") + self.eps)
        return masked_mean(preds, labels, self.kind)
