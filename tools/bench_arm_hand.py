#!/usr/bin/env python
"""The reference's arm + hand task set (examples/arm_hand_iiwa_allegro.py:62-94: FrameTask on the wrist, posture, one
RelativeFrameTask per fingertip measured in the palm) on the 7 + 16-dof model of tests/golden/models/arm_hand.json at 65 536
instances: default dispatch (two-row build of the row kernel) against the wavefront kernel.

    python tools/bench_arm_hand.py [batch]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import torch

    from mink_amd import _native as nat
    from mink_amd.api_specs import configuration_limit_desc
    from mink_amd.flatmodel import FlatModel

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    m = FlatModel.load(os.path.join(REPO, "tests", "golden", "models", "arm_hand.json"))
    nm = nat.NativeModel(m)
    site = lambda s: m.name2id("site", s)
    fts = [{"frame_type": "site", "frame_id": site("attachment_site"), "cost": [1.0] * 6, "gain": 1.0, "lm_damping": 1.0}]
    for t in ("ff_tip", "mf_tip", "rf_tip", "th_tip"):
        fts.append({"frame_type": "site", "frame_id": site(t), "cost": [1.0, 1.0, 1.0, 0.0, 0.0, 0.0], "gain": 1.0, "lm_damping": 1.0,
                    "root_type": "body", "root_id": m.name2id("body", "palm")})
    prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 5e-2}], configuration_limits=[configuration_limit_desc(m)],
                             max_batch=B)
    rng = np.random.default_rng(0)
    home = m.key_qpos[0]
    q = home + rng.normal(scale=0.2, size=(B, m.nq))
    q2 = q + rng.normal(scale=0.15, size=q.shape)
    dummy = np.zeros((B, len(fts), 7)); dummy[:, :, 0] = 1.0
    _, _, t = prob.solve(q2, dummy, home[None, :], None, 1.0, 1.0, taps=["frame_pose"], solve_qp=False)
    tg = t["frame_pose"]                                    # (relative tasks: the tap is the frame in its root frame)
    dev = torch.device("cuda", 0)
    qd, tgd, pt = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (q, tg, home[None, :].copy()))
    v = torch.empty((B, m.nv), dtype=torch.float64, device=dev)
    st = torch.empty((B,), dtype=torch.int32, device=dev)
    res = {}
    for kw in ({}, {"wave_kernel": True}):
        for _ in range(3):
            prob.solve(qd, tgd, pt, None, 1e-2, 1e-3, out=v, status_out=st, **kw)
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); prob.solve(qd, tgd, pt, None, 1e-2, 1e-3, out=v, status_out=st, **kw); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res[prob.last_kernel()] = v.cpu().numpy().copy()
        print("arm + hand, B = %d: %-30s %.3f ms  %.1f M solves/s  failed %d" % (B, prob.last_kernel(), np.median(ts), B / np.median(ts) / 1e3,
                                                                         int(((st.cpu().numpy() & ~1) != 0).sum())))
    a, b = list(res.values())
    print("max rel difference of the two kernels' velocities: %.1e" % (np.abs(a - b).max() / max(1.0, np.abs(b).max())))


if __name__ == "__main__":
    main()
