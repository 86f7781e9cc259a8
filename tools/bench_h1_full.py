import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mink_amd import _native as nat, workloads
from mink_amd.api_specs import configuration_limit_desc, velocity_limit_desc
model = workloads.load_robot("h1"); nm = nat.NativeModel(model)
B = 65536
fd = workloads._frame_desc
fts = [fd(model, "pelvis", "body", 0.0, 10.0)] + [fd(model, s, "site", 200.0, 10.0, 1.0) for s in ("left_foot", "right_foot")] + [fd(model, s, "site", 200.0, 0.0, 1.0) for s in ("left_wrist", "right_wrist")]
for com in (True, False):
    kw = dict(frame_tasks=fts, posture_tasks=[{"cost": 1.0}], configuration_limits=[configuration_limit_desc(model)],
              velocity_limits=[velocity_limit_desc(model, workloads._hinge_velocities(model))], max_batch=B)
    if com: kw["com_tasks"] = [{"cost": 200.0}]
    prob = nat.NativeProblem(nm, **kw)
    rng = np.random.default_rng(0)
    base = model.key_qpos[model.name2id("key", "stand")]
    q, tg = workloads.make_batch(model, nm, prob, rng, B, base_q=base)
    dev = torch.device("cuda", 0)
    qd, tgd = torch.from_numpy(q).to(dev), torch.from_numpy(tg).to(dev)
    pt = torch.from_numpy(base[None, :].copy()).to(dev)
    ct = torch.zeros((1, 3), dtype=torch.float64, device=dev) if com else None
    if com: ct[0, 2] = 0.9
    v = torch.empty((B, model.nv), dtype=torch.float64, device=dev); st = torch.empty((B,), dtype=torch.int32, device=dev)
    for _ in range(3): prob.solve(qd, tgd, pt, ct, 5e-3, 1e-1, out=v, status_out=st)
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prob.solve(qd, tgd, pt, ct, 5e-3, 1e-1, out=v, status_out=st); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print("H1 %s: %s %.3f ms %.1f M/s failed %d" % ("full example (ComTask)" if com else "without ComTask", prob.last_kernel(), np.median(ts), B / np.median(ts) / 1e3, int(((st.cpu().numpy() & ~1) != 0).sum())))
