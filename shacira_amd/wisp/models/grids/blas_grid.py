"""``BLASGrid``: the abstract feature-grid interface the neural fields and tracers program against
(reference wisp/models/grids/blas_grid.py:15-73): ``interpolate`` plus ``raymarch/raytrace/query`` forwarded to
the acceleration structure in ``self.blas``."""
from abc import ABC, abstractmethod
from typing import Any, Dict, Set, Type

from ...accelstructs import BaseAS
from ...core import WispModule


class BLASGrid(WispModule, ABC):
    def __init__(self, blas: BaseAS):
        super().__init__()
        self.blas = blas

    def raymarch(self, *args, **kwargs):
        return self.blas.raymarch(*args, **kwargs)

    def raytrace(self, *args, **kwargs):
        return self.blas.raytrace(*args, **kwargs)

    def query(self, *args, **kwargs):
        return self.blas.query(*args, **kwargs)

    @abstractmethod
    def interpolate(self, coords, lod_idx):
        raise NotImplementedError("A BLASGrid should implement the interpolation functionality according to "
                                  "the grid structure.")

    def supported_blas(self) -> Set[Type[BaseAS]]:
        return set()

    def public_properties(self) -> Dict[str, Any]:
        return {"Acceleration Structure": self.blas}
