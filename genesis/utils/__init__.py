"""genesis.utils -- only the geometry helpers Go2Env imports."""
from . import geom  # noqa: F401
