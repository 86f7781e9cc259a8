"""Model helpers with mink's names and meaning (mink/utils.py:10-174) on a FlatModel.

The reference functions take `mujoco.MjModel` / `MjData`; here the model is the FlatModel and the role of
`MjData` — the forward-kinematics state — is played by the `Configuration` (which evaluates frame poses on
the device).  Only what a caller of the hot path uses to prepare it: configuration vectors, free-joint index
sets (PostureTask / custom costs), geom id sets for `CollisionAvoidanceLimit`, mocap targets.
"""

from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .exceptions import InvalidKeyframe, InvalidMocapBody
from .flatmodel import JNT_FREE, FlatModel, qpos_width


def move_mocap_to_frame(model: FlatModel, configuration, mocap_name: str, frame_name: str, frame_type: str) -> None:
    """mink/utils.py:10-35: put the mocap body `mocap_name` at the current pose of a frame.  `configuration`
    stands where the reference has `data`; a batched configuration uses its first instance (a mocap body is one
    pose of the model)."""
    body = model.name2id("body", mocap_name)
    if body < 0 or int(model.body_mocapid[body]) == -1:
        raise InvalidMocapBody(mocap_name, model)
    mocap_id = int(model.body_mocapid[body])
    pose = configuration.get_transform_frame_to_world(frame_name, frame_type).wxyz_xyz
    pose = np.asarray(pose, dtype=np.float64).reshape(-1, 7)[0]
    model.mocap_pos[mocap_id] = pose[4:]
    model.mocap_quat[mocap_id] = pose[:4]


def get_freejoint_dims(model: FlatModel) -> Tuple[List[int], List[int]]:
    """mink/utils.py:38-56: configuration and tangent indices of every free joint."""
    q_ids: List[int] = []
    v_ids: List[int] = []
    for j in range(model.njnt):
        if model.jnt_type[j] == JNT_FREE:
            qadr, vadr = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
            q_ids.extend(range(qadr, qadr + 7))
            v_ids.extend(range(vadr, vadr + 6))
    return q_ids, v_ids


def custom_configuration_vector(model: FlatModel, key_name: Optional[str] = None, **kwargs) -> np.ndarray:
    """mink/utils.py:59-98: qpos0 (or a keyframe) with the named joints set to the given values."""
    if key_name is not None:
        key_id = model.name2id("key", key_name)
        if key_id == -1:
            raise InvalidKeyframe(key_name, model)
        q = np.array(model.key_qpos[key_id], dtype=np.float64)
    else:
        q = np.array(model.qpos0, dtype=np.float64)
    for name, value in kwargs.items():
        jid = model.name2id("joint", name)
        if jid < 0:
            raise KeyError(f"Invalid name '{name}'. Valid names: {[n for n in model.jnt_names if n]}")
        jnt_dim = qpos_width(int(model.jnt_type[jid]))
        qid = int(model.jnt_qposadr[jid])
        value = np.atleast_1d(value)
        if value.shape != (jnt_dim,):
            raise ValueError(f"Joint {name} should have a qpos value of {jnt_dim,} but got {value.shape}")
        q[qid:qid + jnt_dim] = value
    return q


def get_body_body_ids(model: FlatModel, body_id: int) -> List[int]:
    """mink/utils.py:101-116: immediate children of a body."""
    return [i for i in range(model.nbody) if model.body_parentid[i] == body_id and body_id != i]


def get_subtree_body_ids(model: FlatModel, body_id: int) -> List[int]:
    """mink/utils.py:119-135: every body of the subtree rooted at `body_id` (the reference's stack order)."""
    body_ids: List[int] = []
    stack = [body_id]
    while stack:
        body_id = stack.pop()
        body_ids.append(body_id)
        stack += get_body_body_ids(model, body_id)
    return body_ids


def get_body_geom_ids(model: FlatModel, body_id: int) -> List[int]:
    """mink/utils.py:138-152: geoms attached directly to a body."""
    start = int(model.body_geomadr[body_id])
    return list(range(start, start + int(model.body_geomnum[body_id]))) if start >= 0 else []


def get_subtree_geom_ids(model: FlatModel, body_id: int) -> List[int]:
    """mink/utils.py:155-174: geoms of the subtree rooted at `body_id` (the reference's stack order)."""
    geom_ids: List[int] = []
    stack = [body_id]
    while stack:
        body_id = stack.pop()
        geom_ids.extend(get_body_geom_ids(model, body_id))
        stack += get_body_body_ids(model, body_id)
    return geom_ids
