"""mink/constants.py:3-34 equivalents."""
from .flatmodel import dof_width, qpos_width  # noqa: F401

SUPPORTED_FRAMES = ("body", "geom", "site")

# mink/constants.py:3-24.  The reference maps a frame type to MuJoCo's object enum, to the mj_jac* function and to
# the MjData attribute names; the values below are MuJoCo's own (mjtObj: body 1, geom 5, site 6; function and
# attribute names as strings — there is no mujoco module behind this package).
FRAME_TO_ENUM = {"body": 1, "geom": 5, "site": 6}
FRAME_TO_JAC_FUNC = {"body": "mj_jacBody", "geom": "mj_jacGeom", "site": "mj_jacSite"}
FRAME_TO_POS_ATTR = {"body": "xpos", "geom": "geom_xpos", "site": "site_xpos"}
FRAME_TO_XMAT_ATTR = {"body": "xmat", "geom": "geom_xmat", "site": "site_xmat"}
