"""Exception types of the mink API surface (same names and messages as the reference:
mink/exceptions.py:8-98, mink/tasks/exceptions.py:6-27, mink/limits/exceptions.py:6)."""

from typing import Sequence


class MinkError(Exception):
    """Base class for Mink exceptions."""


class UnsupportedFrame(MinkError):
    def __init__(self, frame_type: str, supported_types: Sequence[str]):
        super().__init__(f"{frame_type} is not supported."
                         f"Supported frame types are: {supported_types}")


class InvalidFrame(MinkError):
    def __init__(self, frame_name: str, frame_type: str, model, reason: str = None):
        names = {"body": model.body_names, "site": model.site_names, "geom": model.geom_names}[frame_type]
        if reason is not None:
            super().__init__(f"{frame_type} '{frame_name}' cannot be used as a frame: {reason}")
            return
        super().__init__(f"{frame_type} '{frame_name}' does not exist in the model. "
                         f"Available {frame_type} names: {list(names)}")


class InvalidKeyframe(MinkError):
    def __init__(self, keyframe_name: str, model):
        super().__init__(f"Keyframe {keyframe_name} does not exist in the model. "
                         f"Available keyframe names: {list(model.key_names)}")


class InvalidMocapBody(MinkError):
    def __init__(self, mocap_name: str, model):
        names = [model.body_names[i] for i in range(model.nbody) if model.body_mocapid[i] != -1]
        super().__init__(f"Body '{mocap_name}' is not a mocap body. Available mocap bodies: {names}")


class NotWithinConfigurationLimits(MinkError):
    def __init__(self, joint_id: int, value: float, lower: float, upper: float, model):
        super().__init__(f"Joint {joint_id} ({model.jnt_names[joint_id]}) violates configuration limits "
                         f"{lower} <= {value} <= {upper}")


class TaskDefinitionError(MinkError):
    """Exception raised when a task definition is ill-formed."""


class TargetNotSet(MinkError):
    def __init__(self, cls_name: str):
        super().__init__(f"No target set for {cls_name}")


class InvalidTarget(MinkError):
    """Exception raised when the target is invalid."""


class InvalidGain(MinkError):
    """Exception raised when the gain is outside the valid range."""


class InvalidDamping(MinkError):
    """Exception raised when the damping is outside the valid range."""


class LimitDefinitionError(MinkError):
    """Exception raised when a limit definition is ill-formed."""


class SolverError(MinkError):
    """A batch instance could not be solved (the reference fails its `assert dq is not None`,
    mink/solve_ik.py:103, or qpsolvers raises ProblemError)."""
