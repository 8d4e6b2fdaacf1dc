"""``WispModule``: nn.Module plus ``name()`` / ``public_properties()`` (reference wisp/core/wisp_module.py:14-46)."""
from abc import ABC, abstractmethod
from typing import Any, Dict

import torch.nn as nn


class WispModule(nn.Module, ABC):
    def __init__(self):
        super().__init__()

    def name(self) -> str:
        return type(self).__name__

    @abstractmethod
    def public_properties(self) -> Dict[str, Any]:
        raise NotImplementedError("Wisp modules should implement the `public_properties` method")


class Rays:
    """Ray batch: ``origins`` / ``dirs`` [N, 3] and the scalar marching interval (reference wisp/core/rays.py:18-35;
    only what the tracer and the acceleration structure read)."""

    def __init__(self, origins, dirs, dist_min=0.0, dist_max=float("inf")):
        self.origins, self.dirs, self.dist_min, self.dist_max = origins, dirs, dist_min, dist_max

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def __len__(self):
        return self.origins.shape[0]

    def __getitem__(self, idx):
        return Rays(self.origins[idx], self.dirs[idx], self.dist_min, self.dist_max)

    def to(self, *args, **kwargs):
        return Rays(self.origins.to(*args, **kwargs), self.dirs.to(*args, **kwargs), self.dist_min, self.dist_max)


class RenderBuffer:
    """The channels the packed tracer fills (reference wisp/core/render_buffer.py; plain attribute bag here)."""

    def __init__(self, rgb=None, alpha=None, depth=None, hit=None, **extra):
        self.rgb, self.alpha, self.depth, self.hit = rgb, alpha, depth, hit
        for k, v in extra.items():
            setattr(self, k, v)
