"""IWAE negative log-likelihood, mirroring evaluate.py:111-123 (the NLL part of `--mode test`).
FID / precision-recall / PPL need pretrained Inception/VGG weights fetched from the network and are
out of scope."""
from __future__ import annotations

import math

import torch

from .util import Metric, ModelEvaluation


def batch_neg_log_likelihood(model, batch, n_attempts: int = 10, eps_lists=None) -> torch.Tensor:
    """evaluate.py:113-122 for ONE batch: -mean_b(logsumexp_k(log p(x|z_k) + log p(z_k) - log q(z_k|x)) - log k).
    eps_lists: optional k lists of per-group N(0,1) noise (one list per importance sample) instead of the
    device RNG, so that the bound can be compared with an external implementation on identical noise."""
    if eps_lists is not None:
        n_attempts = len(eps_lists)
    logs = []
    for a in range(n_attempts):
        reconstruction, _, log_p, log_q = model(batch, nll=True,
                                                eps_list=None if eps_lists is None else eps_lists[a])
        # 28x28 crop of the zero-padded MNIST image (Q5); RGB data sets are not padded
        recon = model.calculate_recon_loss(batch, reconstruction, crop_output=model.head == "bernoulli")
        logs.append(-recon - log_q + log_p)
    return -(torch.logsumexp(torch.stack(logs), dim=0) - math.log(float(n_attempts))).mean()


def neg_log_likelihood(model, test_data, n_attempts: int = 10, eps_lists=None) -> Metric:
    """eps_lists: optional, one entry per batch of `test_data`, each a list of k per-group noise lists."""
    nlls = []
    for bi, (batch, _) in enumerate(test_data):
        nlls.append(float(batch_neg_log_likelihood(model, batch, n_attempts,
                                                   None if eps_lists is None else eps_lists[bi])))
    return Metric.from_list(nlls)     # mean +- std ACROSS batches, as the reference reports it

def save_samples_to_tensorboard(epoch, model, image_logger):
    """evaluate.py:15-21: three samples per temperature as image summaries `t=<temperature>`."""
    for temperature in [0.7, 0.8, 0.9, 1.0]:
        images, *_ = model.sample(temperature=temperature, n_samples=3)
        for i in range(images.shape[0]):
            image_logger.add_image(f"t={temperature:.1f}/image/{i}", images[i], epoch)


def reconstruction_comparison(model, batch: torch.Tensor) -> torch.Tensor:
    """evaluate.py:27-43: the first three test images next to their reconstructions (mean of the output
    distribution), laid out side by side: [3, H, 2W, C] in [0, 1]."""
    batch = batch[:3]                                    # "Tensorboard can only display 3 images"
    logits, *_ = model(batch)
    if model.head == "dmol":
        from . import ops
        # the mixture has no closed-form mean worth showing: draw at u = 1/2 (the component medians)
        half = torch.full(logits.shape[:3] + (model.num_mixture_dec,), 0.5, device=logits.device)
        images = ops.dmol_sample(logits.float().contiguous(), model.num_mixture_dec, 1.0, half,
                                 torch.full(logits.shape[:3] + (3,), 0.5, device=logits.device))
    else:
        images = torch.sigmoid(logits.float())           # Bernoulli(logits).mean()
    return torch.cat((batch.to(images.device, torch.float32), images), dim=2)


def save_reconstructions_to_tensorboard(epoch, model, test_data, image_logger):
    """evaluate.py:24-45 (the reference takes the first batch of a 10-element shuffle buffer; here the
    first batch, deterministic)."""
    batch = next(iter(test_data))[0]
    comparison = reconstruction_comparison(model, batch)
    for i in range(comparison.shape[0]):
        image_logger.add_image(f"test_reconstruction/image/{i}", comparison[i], epoch)
    return comparison


def evaluate_model(epoch, model, test_data, n_attempts: int = 10, **_) -> ModelEvaluation:
    return ModelEvaluation(nll=neg_log_likelihood(model, test_data, n_attempts=n_attempts))
