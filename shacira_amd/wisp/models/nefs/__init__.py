from .nerf import NeuralRadianceField

__all__ = ["NeuralRadianceField"]
