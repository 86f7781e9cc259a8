"""Model breadth: slide / ball / free joints and several joints in one body,
GPU vs oracle on small hand-written MJCF models (cf. reference tests/test_velocity_limit.py:65-89,
tests/test_configuration_limit.py:57-121 which use inline MJCF for the same purpose)."""

import numpy as np
import pytest

import mink_amd as mink
from mink_amd import _native as nat
from oracle import ik as oik
from oracle import lie as olie

pytestmark = pytest.mark.gpu

MIXED = """
<mujoco>
  <compiler angle="radian"/>
  <worldbody>
    <body name="b1" pos="0 0 0.1">
      <inertial pos="0 0 0.05" mass="1" diaginertia="1 1 1"/>
      <joint name="hinge" type="hinge" axis="0 1 0" range="-1.2 1.2" pos="0 0 0.02"/>
      <body name="b2" pos="0.1 0 .3" quat="0.9 0.1 0 0.4">
        <inertial pos="0 0.02 0" mass="0.7" diaginertia="1 1 1"/>
        <joint name="ball" type="ball" pos="0.01 0 0"/>
        <body name="b3" pos="0 0.05 .3">
          <inertial pos="0.03 0 0" mass="0.4" diaginertia="1 1 1"/>
          <joint name="slide" type="slide" axis="1 0.2 0" range="-0.2 0.3"/>
          <site name="tip" pos="0.02 0.01 0.1" quat="0.8 0 0.6 0"/>
          <body name="b4" pos="0 0 .2">
            <inertial pos="0 0 0.1" mass="0.3" diaginertia="1 1 1"/>
            <joint name="px" type="slide" axis="1 0 0"/>
            <joint name="py" type="slide" axis="0 1 0"/>
            <joint name="yaw" type="hinge" axis="0 0 1" pos="0.01 0.02 0"/>
            <joint name="pitch" type="hinge" axis="0 1 0" range="-1 1"/>
            <site name="multi" pos="0.05 0 0.05"/>
          </body>
        </body>
      </body>
    </body>
    <body name="floating" pos="1 0 0.5" quat="0.7 0.1 0.2 0.3">
      <inertial pos="0.01 0.02 0.03" mass="2" diaginertia="1 1 1"/>
      <freejoint name="free"/>
      <site name="fs" pos="0.1 0 0" quat="0.5 0.5 0.5 0.5"/>
      <body name="arm" pos="0 0 0.2">
        <inertial pos="0 0 0.1" mass="0.5" diaginertia="1 1 1"/>
        <joint name="elbow" type="hinge" axis="1 0 0" range="-2 2"/>
        <site name="hand" pos="0 0 0.25"/>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


def _rand_q(m, rng):
    q = np.array(m.qpos0)
    for j in range(m.njnt):
        a, t = int(m.jnt_qposadr[j]), int(m.jnt_type[j])
        if t == 0:
            q[a:a + 3] += rng.normal(scale=0.2, size=3)
            q[a + 3:a + 7] = olie.so3_exp(rng.normal(size=3))
        elif t == 1:
            q[a:a + 4] = olie.so3_exp(rng.normal(size=3))
        else:
            lo, hi = m.jnt_range[j] if m.jnt_limited[j] else (-1.0, 1.0)
            q[a] = rng.uniform(lo, hi)
    return q


def test_mixed_joint_model_vs_oracle():
    m = mink.loads_mjcf(MIXED)
    assert (m.nq, m.nv) == (1 + 4 + 1 + 4 + 7 + 1, 1 + 3 + 1 + 4 + 6 + 1)
    rng = np.random.default_rng(0)
    B = 24
    q = np.stack([_rand_q(m, rng) for _ in range(B)])
    cfg = mink.Configuration(m, q)
    names = [("tip", "site"), ("multi", "site"), ("fs", "site"), ("hand", "site"), ("b2", "body")]
    tasks = []
    tgt_cfg = mink.Configuration(m, cfg.integrate(rng.normal(scale=0.2, size=(B, m.nv)), 1.0))
    for n, t in names:
        ft = mink.FrameTask(n, t, position_cost=[1.0, 2.0, 0.5], orientation_cost=0.3, gain=0.7, lm_damping=0.5)
        ft.set_target(tgt_cfg.get_transform_frame_to_world(n, t))
        tasks.append(ft)
    post = mink.PostureTask(m, cost=np.linspace(0.1, 0.5, m.nv), gain=0.5)
    post.set_target(_rand_q(m, rng))
    com = mink.ComTask(cost=[3.0, 2.0, 1.0])
    com.set_target(np.array([0.1, 0.0, 0.4]))
    lims = [mink.ConfigurationLimit(m, gain=0.8),
            mink.VelocityLimit(m, {"ball": (0.5, 0.6, 0.7), "slide": 0.4, "yaw": 1.0, "elbow": 0.9})]
    dt, damping = 1e-2, 1e-4
    v = mink.solve_ik(cfg, tasks + [post, com], dt, "mi355x", damping, limits=lims)

    def spec(i):
        ts = [oik.FrameTaskSpec(m.name2id(t, n), t, ft.cost, ft.transform_target_to_world.wxyz_xyz[i], 0.7, 0.5)
              for (n, t), ft in zip(names, tasks)]
        ts += [oik.PostureTaskSpec(post.cost, post.target_q, 0.5), oik.ComTaskSpec(com.cost, com.target_com)]
        ls = [oik.ConfigurationLimitSpec(0.8), oik.VelocityLimitSpec(lims[1].indices, lims[1].limit)]
        return ts, ls

    # intermediates of every task, then the solution
    for i in range(B):
        ts, ls = spec(i)
        o = oik.Configuration(m, q[i])
        for k, (task, ot) in enumerate(zip(tasks + [post, com], ts)):
            e_ref, J_ref = oik.task_error_jacobian(o, ot)
            if i < 4:
                one = mink.Configuration(m, q[i])
                t1 = task
                if isinstance(task, mink.FrameTask):
                    t1 = mink.FrameTask(task.frame_name, task.frame_type, task.cost[:3], task.cost[3:], 0.7, 0.5)
                    t1.set_target(mink.SE3(task.transform_target_to_world.wxyz_xyz[i]))
                np.testing.assert_allclose(t1.compute_error(one), e_ref, atol=1e-12)
                np.testing.assert_allclose(t1.compute_jacobian(one), J_ref, atol=1e-10)
        v_ref = oik.solve_ik(m, o, ts, dt, damping, ls)
        np.testing.assert_allclose(v[i], v_ref, rtol=0, atol=1e-8 * max(1.0, np.abs(v_ref).max()))
    # integrate with ball + free + multi-joint bodies
    qn = cfg.integrate(v, dt)
    for i in range(0, B, 5):
        np.testing.assert_allclose(qn[i], oik.Configuration(m, q[i]).integrate(v[i], dt), atol=1e-15)
