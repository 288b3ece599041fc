# Mirrors nvdiffrast/torch/__init__.py:9-10 of the reference.
from .ops import *  # noqa: F401,F403
from .ops import __all__  # noqa: F401
