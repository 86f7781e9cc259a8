"""Per-phase shader-clock breakdown of ik_solve_kernel (cycles tap).  GPU only.
    python tools/phase_profile.py [B] [config=g1_c3] [--direct]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mink_amd import _native as nat, workloads

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 65536
config = args[1] if len(args) > 1 else "g1_c3"
model = workloads.load_robot(workloads.BENCH_CONFIGS[config]["robot"])
nm = nat.NativeModel(model)
prob, dt, damping = workloads.bench_config(config, model, nm, B)
q, tg, stand_t, com_t = workloads.bench_batch(config, model, nm, prob, np.random.default_rng(0), B)
stand = stand_t[0] if stand_t is not None else model.qpos0
for _ in range(2):
    v, st, t = prob.solve(q, tg, stand[None, :], com_t, dt, damping, taps=["cycles", "qp_iters"], wave_kernel=True)
c = t["cycles"].astype(np.int64)
d = np.diff(c[:, :8], axis=1)
names = ["load+FK", "axes/dof/com", "task lanes", "posture+J cols", "limits+coll", "build T + phase 0", "GI"]
print("kernel", prob.last_kernel())
tot = (c[:, 7] - c[:, 0])
print("launch", prob.launch_info(B))
print("per-problem wave cycles: mean %.0f  p50 %.0f  p99 %.0f  max %d" % (tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max()))
for k, n in enumerate(names):
    print("  %-16s mean %8.0f  (%4.1f%%)" % (n, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))
print("    %-18s mean %8.0f   (then low-rank S block / rhs: %.0f)" % ("J columns", (c[:, 14] - c[:, 3]).mean(), (c[:, 4] - c[:, 14]).mean()))
sub = ["phase-0 publish", "phase-0 rcp+pivot", "GI select", "GI publish", "GI ratio test", "GI pivot"]
if "_r" in prob.last_kernel():      # low-rank start: slots 0-3 are the stages of wood_start and the rank-1 updates that follow
    sub = ["low-rank: J rows", "low-rank: S, w", "low-rank: elimination", "low-rank: rank-1 dof block (+GI publish)", "GI ratio test", "GI pivot"]
for k, n in enumerate(sub):
    print("    %-18s mean %8.0f" % (n, c[:, 8 + k].mean()))
it = t["qp_iters"]
print("GI: selections mean %.1f, loop iterations %.1f, rank-1 pivots %.1f; cycles per loop iteration %.0f" % (
    it.mean(), t["qp_loops"].mean(), t["qp_pivots"].mean(), d[:, 6].mean() / max(t["qp_loops"].mean(), 1e-9)))
if "--direct" in sys.argv:
    v, st, t = prob.solve(q, tg, stand[None, :], com_t, dt, damping, taps=["cycles", "qp_iters"], direct_qp=True)
    c = t["cycles"].astype(np.int64); d = np.diff(c[:, :8], axis=1); tot = c[:, 7] - c[:, 0]
    print("kernel", prob.last_kernel(), "mean cycles %.0f" % tot.mean())
    for k, n in enumerate(names):
        print("  %-16s mean %8.0f  (%4.1f%%)" % (n, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean()))
