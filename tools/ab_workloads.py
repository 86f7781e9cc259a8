import os, sys, numpy as np, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mink_amd import _native as nat, workloads
names = sys.argv[1:] or ["g1_c3"]
for name in names:
    B = workloads.BENCH_CONFIGS[name]["batch"]
    model = workloads.load_bench_robot(name); nm = nat.NativeModel(model)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    q, tg, pt, ct = workloads.bench_batch(name, model, nm, prob, np.random.default_rng(0), B)
    dense = workloads.bench_dense(name, model, nm, q, np.random.default_rng(1))
    dev = torch.device("cuda", 0); to = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    q, tg, pt, ct = to(q), to(tg), to(pt), to(ct)
    dd = None if dense is None else {k: to(v) for k, v in dense.items()}
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev); st = torch.empty((B,), dtype=torch.int32, device=dev)
    for _ in range(5): prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dd)
    ev = []
    for _ in range(40):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prob.solve(q, tg, pt, ct, dt, damping, out=v, status_out=st, dense=dd); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(os.environ.get("MKH_LIB_TAG", ""), name, prob.last_kernel(), "median %.4f min %.4f ms" % (statistics.median(ms), ms[0]), "checksum %.12e" % float(v.abs().sum()))
    prob.close()
