"""IWAE negative log-likelihood, mirroring evaluate.py:111-123 (the NLL part of `--mode test`).
FID / precision-recall / PPL need pretrained Inception/VGG weights fetched from the network and are
out of scope."""
from __future__ import annotations

import math

import torch

from .util import Metric, ModelEvaluation


def neg_log_likelihood(model, test_data, n_attempts: int = 10) -> Metric:
    nlls = []
    for batch, _ in test_data:
        logs = []
        for _ in range(n_attempts):
            reconstruction, _, log_p, log_q = model(batch, nll=True)
            # 28x28 crop of the zero-padded MNIST image (Q5); RGB data sets are not padded
            recon = model.calculate_recon_loss(batch, reconstruction, crop_output=model.head == "bernoulli")
            logs.append(-recon - log_q + log_p)
        nll = -(torch.logsumexp(torch.stack(logs), dim=0) - math.log(float(n_attempts))).mean()
        nlls.append(float(nll))
    return Metric.from_list(nlls)     # mean +- std ACROSS batches, as the reference reports it


def evaluate_model(epoch, model, test_data, n_attempts: int = 10, **_) -> ModelEvaluation:
    return ModelEvaluation(nll=neg_log_likelihood(model, test_data, n_attempts=n_attempts))
