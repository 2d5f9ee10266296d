"""Learned loss weighting of EPMF -- call surface of pc_processor/loss/multi_task_loss.py:5-19.

    total = sum_i  L_i / (2 sigma_i^2) + log(sigma_i^2 + 1)

A handful of scalars: plain device-agnostic torch ops (no kernel to write)."""
import torch
import torch.nn as nn


class MultiTaskLoss(nn.Module):
    def __init__(self, n_losses, sigma=None):
        super().__init__()
        if sigma is not None:
            self.sigma = nn.Parameter(torch.Tensor(sigma))
        else:
            self.sigma = nn.Parameter(torch.ones(n_losses) / n_losses)

    def forward(self, losses):
        total_loss = 0
        for i, loss in enumerate(losses):
            total_loss = total_loss + (loss / (2.0 * self.sigma[i].pow(2)) + (self.sigma[i].pow(2) + 1.0).log())
        return total_loss
