"""A BASELINE-sized slice of config 3 from the REAL reference: 2 048 G1 instances (q, 4 frame targets) → v.

Run in the build container only (needs /root/reference; ≈30 s at the reference's ≈77 solves/s):

    python tests/golden/make_golden_big.py

Same set-up, sampling and stubs as make_golden.py (mink's own Python runs, mujoco / qpsolvers are oracle/stubs).
The small fixtures hold 12–32 instances each; this one is large enough that the production kernel
(`ik_solve_kernel_44_32_r44_w3`: low-rank start + cold-start refinement, whose intermediates cannot be tapped) runs
several persistent rounds per wavefront and reaches its ticket-counter tail against the imported reference itself.
Only inputs and v are stored (1.9 MB)."""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up sys.path: stubs, /root/reference, repo)

mink, mujoco = mg.mink, mg.mujoco


def main(n=2048):
    rng = np.random.default_rng(2024)
    m = mujoco.MjModel.from_xml_path(mg.ROBOTS["g1"])
    stand = m.key_qpos[m.key("stand").id]
    feet = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=10.0, lm_damping=1.0)
            for s in ("left_foot", "right_foot")]
    hands = [mink.FrameTask(s, "site", position_cost=200.0, orientation_cost=0.0, lm_damping=1.0)
             for s in ("left_palm", "right_palm")]
    pt = mink.PostureTask(m, cost=1.0)
    pt.set_target(stand)
    vel = {m.jnt_names[j]: np.pi for j in range(m.njnt) if m.jnt_type[j] != 0}
    lims = [mink.ConfigurationLimit(m), mink.VelocityLimit(m, vel)]
    tasks, fts = feet + hands + [pt], feet + hands
    qb = mg.sample_q(m, rng, n, base_q=stand)
    tg_all, v_all = [], []
    t0 = time.time()
    for i, q in enumerate(qb):
        sig = 1e-4 if (i % 8 == 7) else 0.15            # small-angle stress sub-stream, as in make_golden.run_config
        cfg_t = mink.Configuration(m, mg.perturbed(m, q, rng, sig))
        tg = []
        for t in fts:
            T = cfg_t.get_transform_frame_to_world(t.frame_name, t.frame_type)
            t.set_target(T)
            tg.append(T.wxyz_xyz.copy())
        cfg = mink.Configuration(m, q)
        v_all.append(mink.solve_ik(cfg, tasks, 5e-3, "quadprog", 1e-1, limits=lims))
        tg_all.append(np.array(tg))
    out = dict(q=qb, frame_targets=np.array(tg_all), v=np.array(v_all), posture_target=stand.copy(),
               dt=np.array(5e-3), damping=np.array(1e-1))
    np.savez_compressed(os.path.join(HERE, "ik_g1_c3_big.npz"), **out)
    print("ik_g1_c3_big.npz", {k: v.shape for k, v in out.items()}, f"{time.time() - t0:.1f} s",
          "max|v|", float(np.abs(out["v"]).max()))


if __name__ == "__main__":
    main()
