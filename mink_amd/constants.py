"""mink/constants.py:3-34 equivalents."""
from .flatmodel import dof_width, qpos_width  # noqa: F401

SUPPORTED_FRAMES = ("body", "geom", "site")
