"""Entropy models of the latent path: the factorised `BitEstimator` and its per-layer `Bitparm`."""
from .bit_estimator import BitEstimator, Bitparm

__all__ = ["BitEstimator", "Bitparm"]
