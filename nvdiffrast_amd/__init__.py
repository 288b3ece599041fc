"""nvdiffrast_amd -- MI355X (gfx950) implementation of the nvdiffrast.torch hot path.

Use ``import nvdiffrast_amd.torch as dr`` exactly like ``import nvdiffrast.torch as dr``.
"""
__version__ = "0.1.0"
