"""Host-side mirror of the reference interface for the hash-grid / latent path (package names follow
``/root/reference/wisp`` so callers read the same): ``ops.grid``, ``models.grids``, ``models.latent_decoders``,
``models.prob_models``, ``utils.schedulers``, ``ops.image.metrics``.

``shacira_amd.wisp.install_as_wisp()`` registers these modules under the reference's import names
(``wisp.ops.grid`` ...), so trainers written against the reference import them unchanged.
"""
import importlib
import sys

_ALIASES = [
    "ops", "ops.grid", "ops.image", "ops.image.metrics", "core", "accelstructs", "models", "models.grids",
    "models.grids.blas_grid", "models.grids.hash_grid", "models.grids.latent_grid", "models.latent_decoders",
    "models.latent_decoders.basic_latent_decoder", "models.latent_decoders.hierarchical_latent_decoder",
    "models.latent_decoders.multi_latent_decoder",
    "models.prob_models", "models.prob_models.bit_estimator", "models.decoders", "models.decoders.basic_decoders",
    "models.embedders", "models.nefs", "models.nefs.nerf", "tracers", "tracers.packed_rf_tracer", "utils", "utils.schedulers",
]


def install_as_wisp(force=False):
    """Alias this package as ``wisp`` in ``sys.modules`` (no-op if a real ``wisp`` is already imported)."""
    if "wisp" in sys.modules and not force and sys.modules["wisp"] is not sys.modules[__name__]:
        raise RuntimeError("a different `wisp` package is already imported")
    sys.modules["wisp"] = sys.modules[__name__]
    for name in _ALIASES:
        sys.modules["wisp." + name] = importlib.import_module(__name__ + "." + name)
    return sys.modules["wisp"]
