#!/usr/bin/env python
"""When does the low-rank start pay?  Example robots of the reference with task sets whose row count lies between half and
three quarters of the dof count (where the host's criterion, minkhip.hip `wood_nt`, used to say no below 32 dofs): default
dispatch against MKH_DEBUG_WOOD_ALWAYS=1 (the criterion switched off), same batch, device-resident, median of 20 launches.

    python tools/bench_wood_criterion.py [batch]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

CASES = [  # scene, frames (site / body names), orientation cost of the first two
    ("unitree_h1__scene", ["left_foot", "right_foot", "left_wrist", "right_wrist"], 10.0),       # 18 rows / 25 dofs
    ("unitree_h1__scene", ["left_foot", "right_foot", "left_wrist", "right_wrist"], 0.0),        # 12 / 25
    ("unitree_go1__scene", ["FL", "FR", "RL", "RR"], 0.0),                                        # 12 / 18
    ("boston_dynamics_spot__scene", ["FL", "FR", "HL", "HR"], 0.0),
    ("shadow_hand__scene_left", ["thumb", "first", "middle", "ring", "little"], 0.0),             # 15 / 24
    ("wonik_allegro__scene_left", None, 0.0),
    ("leap_hand__scene_left", None, 0.0),
    # beyond three quarters
    ("unitree_h1__scene", ["left_foot", "right_foot", "left_wrist", "right_wrist"], -10.0),      # orientation on all four: 24 / 25
    ("unitree_go1__scene", ["FL", "FR", "RL", "RR"], 10.0),                                       # 18 / 18
    ("shadow_hand__scene_left", ["thumb", "first", "middle", "ring", "little"], 1.0),             # 21 / 24
]


def main():
    import torch

    import native_configs as nc
    from mink_amd import _native as nat
    from mink_amd import workloads
    from mink_amd.flatmodel import FlatModel

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dev = torch.device("cuda", 0)
    for scene, names, ori in CASES:
        m = FlatModel.load(os.path.join(REPO, "tests", "golden", "models", "all", scene + ".json"))
        nm = nat.NativeModel(m)
        if names is None:
            ids = [i for i, n in enumerate(m.site_names) if n and m.site_bodyid[i] > 0][-4:]
        else:
            ids = [m.name2id("site", n) for n in names]
            if min(ids) < 0:
                print(scene, "sites", [n for n in m.site_names if n]); continue
        fts = [{"frame_type": "site", "frame_id": i, "cost": [1.0] * 3 + [abs(ori) if (k < 2 or ori < 0) else 0.0] * 3, "gain": 1.0, "lm_damping": 1.0}
               for k, i in enumerate(ids)]
        rows = sum(sum(1 for c in f["cost"] if c > 0) for f in fts)
        vidx = [int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] in (2, 3)]
        cells = []
        for force in (False, True):
            if force:
                os.environ["MKH_DEBUG_WOOD_ALWAYS"] = "1"
            else:
                os.environ.pop("MKH_DEBUG_WOOD_ALWAYS", None)
            prob = nat.NativeProblem(nm, frame_tasks=fts, posture_tasks=[{"cost": 1e-2}], configuration_limits=[nc._cfg_limit(m)],
                                     velocity_limits=[{"indices": vidx, "limit": np.full(len(vidx), np.pi)}], max_batch=B)
            if not force:
                q, tg = workloads.make_batch(m, nm, prob, np.random.default_rng(1), B, base_q=m.qpos0)
                qd, tgd = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev)
                pt = torch.from_numpy(m.qpos0[None, :].copy()).to(dev)
            v = torch.empty((B, m.nv), dtype=torch.float64, device=dev)
            st = torch.empty((B,), dtype=torch.int32, device=dev)
            for _ in range(3):
                prob.solve(qd, tgd, pt, None, 5e-3, 1e-2, out=v, status_out=st)
            ts = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                prob.solve(qd, tgd, pt, None, 5e-3, 1e-2, out=v, status_out=st)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            cells.append((prob.last_kernel(), float(np.median(ts)), v.cpu().numpy(), int(((st.cpu().numpy() & ~1) != 0).sum())))
            prob.close()
        (k0, t0, v0, f0), (k1, t1, v1, f1) = cells
        err = np.abs(v0 - v1).max() / max(1.0, np.abs(v0).max())
        print("%-32s nv %2d rows %2d | %-28s %7.3f ms | always: %-28s %7.3f ms  x%.2f  max rel diff %.1e failed %d/%d" %
              (scene, m.nv, rows, k0, t0, k1, t1, t0 / t1, err, f0, f1), flush=True)


if __name__ == "__main__":
    main()
