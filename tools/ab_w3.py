#!/usr/bin/env python
"""A/B of the 3-waves-per-SIMD kernel variants (ik_kernel.h MKH_W3) against the 2-waves maps on one GPU:
kernel time by HIP events, agreement of the two results, and the C oracle on a sample.

    python tools/ab_w3.py [reps]
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import torch

    import native_configs as nc
    import oracle_configs as oc
    from mink_amd import _native as nat
    from mink_amd import workloads
    from oracle import cport

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    for name, B, kw in (("g1_c3", 65536, dict(direct_qp=True)), ("g1_c3", 65536, dict()), ("g1_full", 65536, dict())):
        model = workloads.load_robot("g1")
        nm = nat.NativeModel(model)
        prob, dt, damping = nc.build(name, nm, B)
        base = model.key_qpos[model.name2id("key", "stand")]
        q, tg = workloads.make_batch(model, nm, prob, np.random.default_rng(5), B, base_q=base)
        com = None
        com_np = None
        if prob.n_com:
            _, _, t = prob.solve(q[:4096], tg[:4096], base[None, :], np.zeros((1, 3)), dt, damping, taps=["subtree_com"], solve_qp=False)
            com_np = np.tile(t["subtree_com"][:, None, :] + 0.01, (B // 4096, 1, 1))
            com = torch.from_numpy(com_np).to(dev)
        qd, tgd = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev)
        pt = torch.from_numpy(base[None, :].copy()).to(dev)
        res = {}
        for two in (True, False):
            v = torch.empty((B, model.nv), dtype=torch.float64, device=dev)
            st = torch.empty((B,), dtype=torch.int32, device=dev)
            for _ in range(3):
                prob.solve(qd, tgd, pt, com, dt, damping, out=v, status_out=st, two_waves=two, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                prob.solve(qd, tgd, pt, com, dt, damping, out=v, status_out=st, two_waves=two, **kw)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ms = float(np.median(ts))
            bad = int(((st.cpu().numpy() & ~1) != 0).sum())
            res[two] = v.cpu().numpy()
            li = prob.launch_info(B)
            grid, lds = li["grid"], li["lds_bytes"]
            print("%-8s %-12s %-28s grid %5d lds %6d  %8.3f ms  %7.2f M solves/s  failed %d" %
                  (name, "plain map" if two else "default", prob.last_kernel(), grid, lds, ms, B / ms / 1e3, bad), flush=True)
        d = np.abs(res[True] - res[False]).max()
        print("   max |v(default) - v(2 waves)| = %.3e" % d)
        # C oracle on a sample of the default result
        n = 2048
        if name == "g1_c3":
            m, tasks, limits, dt_o, damp_o = oc.g1_c3(tg[0], base)
            v_c, st_c = cport.CProblem(m, tasks, limits).solve_batch(q[:n], tg[:n], base[None, :], dt_o, damp_o)
            err = np.abs(res[False][:n] - v_c).max(axis=1) / np.maximum(1.0, np.abs(v_c).max(axis=1))
            print("   max rel |v - C oracle| over %d = %.3e" % (n, err.max()))


if __name__ == "__main__":
    main()
