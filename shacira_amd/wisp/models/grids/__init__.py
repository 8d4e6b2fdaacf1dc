from .blas_grid import BLASGrid
from .hash_grid import HashGrid, geometric_resolutions
from .latent_grid import LatentGrid
