"""vts: the VAE train step (LSTM and transformer encoders) of letrucnhi160703/vae-time-series on MI355X (gfx950).

Python face of libvts_hip.so: drop-in modules and functions for the reference's torch call
sites.  All arithmetic runs in the HIP library; torch supplies device memory, streams and
torch.distributed.  Importing works without a GPU; calling any op without one raises.
"""
from . import _lib
from .functional import get_mode, set_mode, set_loss_reduction, set_fused_chains, mlp_chain
from .functional import RecurrenceTimeout, recurrence_status, set_status_checks
from .functional import (gaussian_elbo, gpd_elbo, layer_norm, linear, lstm, mixture_elbo,
                         reparameterize_gaussian, reparameterize_gpd, reparameterize_mixture,
                         mixture_split_elbo, multi_head_attention, dropout_add, dropout_mask,
                         reparameterize_mixture3, mixture3_elbo)
from .modules import (GPDVAE, LSTM, GaussianVAE, LayerNorm, Linear, MixtureVAE, build_model,
                      TransformerEncoderLayer, TransformerMixtureVAE, ThreeComponentVAE)
from .train import Adam, FlatParameters, GraphedTrainStep, train_step, raise_on_recurrence_timeout
from .parallel import DataParallelTrainer, shard_batch
from .windows import WindowSource, create_sliding_windows

__all__ = [
    "LSTM", "Linear", "LayerNorm", "GaussianVAE", "GPDVAE", "MixtureVAE", "build_model",
    "lstm", "linear", "layer_norm", "reparameterize_gaussian", "reparameterize_gpd",
    "reparameterize_mixture", "gaussian_elbo", "gpd_elbo", "mixture_elbo", "mixture_split_elbo",
    "multi_head_attention", "dropout_add", "dropout_mask", "TransformerEncoderLayer", "TransformerMixtureVAE",
    "set_mode", "get_mode", "set_loss_reduction", "Adam", "FlatParameters", "train_step", "GraphedTrainStep", "DataParallelTrainer", "shard_batch",
    "RecurrenceTimeout", "recurrence_status", "set_status_checks", "raise_on_recurrence_timeout",
    "WindowSource", "create_sliding_windows", "ThreeComponentVAE", "reparameterize_mixture3", "mixture3_elbo",
]
