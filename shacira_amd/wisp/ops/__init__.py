"""Operator-level API of the mirror: `grid` (hash-grid autograd Functions) and `image.metrics`."""
