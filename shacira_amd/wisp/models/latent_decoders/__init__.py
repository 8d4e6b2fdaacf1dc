from .basic_latent_decoder import *  # noqa: F401,F403
from .basic_latent_decoder import (DecoderIdentity, DecoderLayer, LatentDecoder, SineScaled, StraightThrough,
                                   StraightThroughFloor, get_dft_matrix)
from .hierarchical_latent_decoder import HierarchicalLatentDecoder
from .multi_latent_decoder import MultiLatentDecoder, MultiLatentDecoderLayer, StraightThroughOneHot
