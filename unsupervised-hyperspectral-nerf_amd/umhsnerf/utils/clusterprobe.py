"""Cosine-similarity soft assignment of rendered spectra to the endmembers (``utils/clusterprobe.py:17-38``).
[R,B]x[B,C] per step: host-side torch ops on the device, not a hot kernel."""
import torch.nn as nn
import torch.nn.functional as F
import torch


class ClusterLookup(nn.Module):
    def __init__(self, dim: int, n_classes: int):
        super().__init__()
        self.n_classes, self.dim = n_classes, dim

    def forward(self, x, alpha, log_probs=False, clusters=None):
        clusters = self.clusters if clusters is None else clusters
        ip = torch.matmul(F.normalize(x, dim=1), F.normalize(clusters, dim=1).t())
        if alpha is None:
            probs = F.one_hot(torch.argmax(ip, dim=1), clusters.shape[0]).to(torch.float32)
        else:
            probs = F.softmax(ip * alpha, dim=1)
        if log_probs:
            return F.log_softmax(ip * alpha, dim=1)
        return ip, probs
