"""Named model configurations: BASELINE.json's five configs in this package's constructor terms.

C1/C2/C3 are the reference's MNIST model (train.py:145-215).  C4/C5 do not exist in the reference
(train.py:219, datasets.py:23-25 and README.md:25-27 leave CIFAR-10 / CelebA as to-do); their shapes
follow the NVAE paper's table of hyper-parameters, expressed with this code base's cells (so the
postprocess tower keeps the reference's dense 5x5 convolutions):

  cifar10   32x32x3, one latent scale of 30 groups at 16x16, 128 initial channels, 2 cells per group,
            one pre/post-process block, mixture of 10 discretised logistics.
  celeba64  64x64x3, three latent scales at 32x32 / 16x16 / 8x8 with 10 / 10 / 20 groups (40 in all;
            the paper's 20 groups at the top scale halved per scale with a floor of 10), 64 initial
            channels, 2 cells per group, one pre/post-process block, mixture of logistics.

`n_groups_per_scale` is bottom-up like the reference's flag (highest resolution first)."""
from __future__ import annotations

from typing import Dict

CONFIGS: Dict[str, dict] = {
    "mnist_c1": dict(n_encoder_channels=32, n_decoder_channels=32, res_cells_per_group=1, n_preprocess_blocks=2,
                     n_preprocess_cells=3, n_latent_per_group=20, n_groups_per_scale=[1, 1],
                     n_postprocess_blocks=2, n_post_process_cells=3, input_hwc=(32, 32, 1), batch=32),
    "mnist_c2": dict(n_encoder_channels=32, n_decoder_channels=32, res_cells_per_group=2, n_preprocess_blocks=2,
                     n_preprocess_cells=3, n_latent_per_group=20, n_groups_per_scale=[5, 10],
                     n_postprocess_blocks=2, n_post_process_cells=3, input_hwc=(32, 32, 1), batch=128),
    "cifar10": dict(n_encoder_channels=128, n_decoder_channels=128, res_cells_per_group=2, n_preprocess_blocks=1,
                    n_preprocess_cells=3, n_latent_per_group=20, n_groups_per_scale=[30],
                    n_postprocess_blocks=1, n_post_process_cells=3, input_hwc=(32, 32, 3), batch=64),
    "celeba64": dict(n_encoder_channels=64, n_decoder_channels=64, res_cells_per_group=2, n_preprocess_blocks=1,
                     n_preprocess_cells=2, n_latent_per_group=20, n_groups_per_scale=[10, 10, 20],
                     n_postprocess_blocks=1, n_post_process_cells=2, input_hwc=(64, 64, 3), batch=32),
}
CONFIGS["mnist_c3"] = dict(CONFIGS["mnist_c2"])      # C2 per GPU, 8 ranks


def build(name: str, batch: int = None, device="cuda:0", dtype=None, sr_lambda=0.01, total_epochs=400,
          n_total_iterations=400 * 417, **kw):
    """Construct the NVAE for a named configuration (weights random-initialised)."""
    import torch
    from .models import NVAE
    c = CONFIGS[name]
    B = batch or c["batch"]
    H, W, C = c["input_hwc"]
    return NVAE(c["n_encoder_channels"], c["n_decoder_channels"], c["res_cells_per_group"],
                c["n_preprocess_blocks"], c["n_preprocess_cells"], c["n_latent_per_group"],
                len(c["n_groups_per_scale"]), c["n_groups_per_scale"], c["n_postprocess_blocks"],
                c["n_post_process_cells"], sr_lambda, 2, total_epochs, n_total_iterations, True, [B, H, W, C],
                device=device, dtype=dtype or torch.bfloat16, **kw)
