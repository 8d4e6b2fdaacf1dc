"""Module-level API of the mirror: grids, latent decoders, entropy models, decoder MLPs, embedders, neural fields."""
