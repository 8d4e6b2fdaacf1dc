from .packed_rf_tracer import PackedRFTracer

__all__ = ["PackedRFTracer"]
