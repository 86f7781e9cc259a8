import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mink_amd import _native as nat, workloads
m = workloads.load_robot("ur5e"); nm = nat.NativeModel(m)
dev = torch.device("cuda", 0)
for B in (64, 256, 1024, 2048, 4096, 8192):
    prob, dt, damping = workloads.bench_config("ur5e_c2", m, nm, B)
    q, tg, pt, ct = workloads.bench_batch("ur5e_c2", m, nm, prob, np.random.default_rng(0), B)
    qd, tgd, ptd = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev), torch.from_numpy(pt).to(dev)
    for wave, lane in ((True, False), (False, True)):
        for _ in range(3): prob.solve(qd, tgd, ptd, None, dt, damping, wave_kernel=wave, lane_kernel=lane)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ts = []
        for _ in range(20):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); prob.solve(qd, tgd, ptd, None, dt, damping, wave_kernel=wave, lane_kernel=lane); e.record(); e.synchronize()
            ts.append(s.elapsed_time(e))
        print("B=%5d %-22s median %.1f us  min %.1f us" % (B, prob.last_kernel(), np.median(ts) * 1e3, min(ts) * 1e3))
