"""Host-side construction of limit descriptors from a FlatModel (setup time).

These restate the *constructors* of the reference's limit classes — the per-solve
arithmetic is on the device:
  ConfigurationLimit.__init__   mink/limits/configuration_limit.py:41-67
  VelocityLimit.__init__        mink/limits/velocity_limit.py:45-69
"""

from __future__ import annotations

from typing import Mapping

import numpy as np

from .flatmodel import JNT_FREE, FlatModel, dof_width, mjMAXVAL, qpos_width


def configuration_limit_desc(model: FlatModel, gain: float = 0.95, min_distance_from_limits: float = 0.0) -> dict:
    index_list = []
    lower = np.full(model.nq, -mjMAXVAL)
    upper = np.full(model.nq, mjMAXVAL)
    for jnt in range(model.njnt):
        jt = int(model.jnt_type[jnt])
        if jt == JNT_FREE or not model.jnt_limited[jnt]:
            continue
        padr = int(model.jnt_qposadr[jnt])
        lower[padr:padr + qpos_width(jt)] = model.jnt_range[jnt, 0] + min_distance_from_limits
        upper[padr:padr + qpos_width(jt)] = model.jnt_range[jnt, 1] - min_distance_from_limits
        va = int(model.jnt_dofadr[jnt])
        index_list.extend(range(va, va + dof_width(jt)))
    return {"gain": float(gain), "lower": lower, "upper": upper,
            "indices": np.array(index_list, dtype=np.int32)}


def velocity_limit_desc(model: FlatModel, velocities: Mapping[str, object]) -> dict:
    idx, lim = [], []
    for name, max_vel in velocities.items():
        jid = model.joint(name).id
        jt = int(model.jnt_type[jid])
        va = int(model.jnt_dofadr[jid])
        mv = np.atleast_1d(np.asarray(max_vel, dtype=np.float64))
        idx.extend(range(va, va + dof_width(jt)))
        lim.extend(mv.tolist())
    return {"indices": np.array(idx, dtype=np.int32), "limit": np.array(lim, dtype=np.float64)}
