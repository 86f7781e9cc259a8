"""mink_amd — MI355X-native batched differential IK behind mink's `solve_ik` / Task / Limit API.

The public names mirror kevinzakka/mink (mink/__init__.py:47-86) for the accelerated path.
"""

from .configuration import Configuration
from .constants import (FRAME_TO_ENUM, FRAME_TO_JAC_FUNC, FRAME_TO_POS_ATTR, FRAME_TO_XMAT_ATTR,
                        SUPPORTED_FRAMES)
from .exceptions import (InvalidDamping, InvalidFrame, InvalidGain, InvalidKeyframe, InvalidMocapBody,
                         InvalidTarget, LimitDefinitionError, MinkError, NotWithinConfigurationLimits,
                         SolverError, TargetNotSet, TaskDefinitionError, UnsupportedFrame)
from .flatmodel import FlatModel
from .lie import SE3, SO3, MatrixLieGroup
from .limits import CollisionAvoidanceLimit, ConfigurationLimit, Constraint, Limit, VelocityLimit
from .mjcf import load_mjcf, loads_mjcf
from .solve_ik import Problem, build_ik, solve_ik, solve_ik_steps
from .tasks import ComTask, DampingTask, FrameTask, Objective, PostureTask, RelativeFrameTask, Task
from .utils import (custom_configuration_vector, get_body_geom_ids, get_freejoint_dims, get_subtree_body_ids,
                    get_subtree_geom_ids, move_mocap_to_frame)
from .workloads import load_robot

__all__ = (
    "ComTask", "Configuration", "build_ik", "solve_ik", "solve_ik_steps", "DampingTask", "FrameTask", "RelativeFrameTask",
    "PostureTask", "Task", "Objective", "ConfigurationLimit", "VelocityLimit", "CollisionAvoidanceLimit",
    "Constraint", "Limit", "SO3", "SE3", "MinkError", "UnsupportedFrame", "InvalidFrame", "InvalidKeyframe",
    "NotWithinConfigurationLimits", "TargetNotSet", "InvalidMocapBody", "SUPPORTED_FRAMES", "FlatModel",
    "load_mjcf", "loads_mjcf", "load_robot", "Problem", "SolverError", "TaskDefinitionError", "InvalidTarget",
    "InvalidGain", "InvalidDamping", "LimitDefinitionError", "MatrixLieGroup", "FRAME_TO_ENUM", "FRAME_TO_JAC_FUNC",
    "FRAME_TO_POS_ATTR", "FRAME_TO_XMAT_ATTR", "custom_configuration_vector", "get_body_geom_ids", "get_freejoint_dims",
    "get_subtree_body_ids", "get_subtree_geom_ids", "move_mocap_to_frame",
)
