import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mink_amd import _native as nat, workloads
for name in ["g1_c3","g1_full","shadow_c4","g1_plugin","g1_coll","aloha_coll","ur5e_coll","ur5e_convex","ur5e_c2"]:
    B = workloads.BENCH_CONFIGS[name]["batch"]
    model = workloads.load_bench_robot(name)
    nm = nat.NativeModel(model)
    prob, dt, damping = workloads.bench_config(name, model, nm, B)
    rng = np.random.default_rng(1)
    q, tg, pt, com = workloads.bench_batch(name, model, nm, prob, rng, B)
    dense = workloads.bench_dense(name, model, nm, q, rng)
    prob.solve(q, tg, pt, com, dt, damping, dense=dense)
    info = prob.launch_info(B)
    lds = info["lds_bytes"]; g = -(-lds // 1280)
    print(name, prob.last_kernel(), info, "granules", g, "by granules", 128 // g if g else None, "by division", (160*1024)//lds if lds else None, "grid/256", info["grid"]/256)
