"""Losses for online (per-time-step) prediction with ``NeuralCDE(return_sequences=True)``: mirrors of the two small
wrappers the reference trains with (/root/reference/experiments/ingredients/metrics.py:26-58).  Plain torch ops on the
model outputs -- nothing here touches the solve."""
import torch
from torch import nn


class TemporalLossWrapper(nn.Module):
    """Applies ``criterion`` over the positions of ``labels`` [N, L, C] that are not NaN (a NaN label marks a time
    step after the series has finished), as metrics.py:26-46 does."""

    def __init__(self, criterion):
        super().__init__()
        assert isinstance(criterion, nn.Module)
        self.criterion = criterion

    def forward(self, preds, labels):
        mask = ~torch.isnan(labels)
        return self.criterion(preds[mask], labels[mask])


class RMSELoss(nn.Module):
    """sqrt(MSE + eps) (metrics.py:49-58)."""

    def __init__(self, eps=1e-6):
        super().__init__()
        self.mse = nn.MSELoss()
        self.eps = eps

    def forward(self, yhat, y):
        return torch.sqrt(self.mse(yhat, y) + self.eps)
