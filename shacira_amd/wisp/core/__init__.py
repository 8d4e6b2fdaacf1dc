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
