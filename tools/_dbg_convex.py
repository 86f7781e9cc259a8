import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import multiprocessing as mp
from mink_amd import _native as nat, workloads
import test_gpu_scale as T
name = "ur5e_convex"
B = 4096
model = workloads.load_bench_robot(name)
nm = nat.NativeModel(model)
prob, dt, damping = workloads.bench_config(name, model, nm, B)
q, tg, pt, _ = workloads.bench_batch(name, model, nm, prob, np.random.default_rng(2024), B)
v, st = prob.solve(q, tg, pt, None, dt, damping)
_, _, t = prob.solve(q, tg, pt, None, dt, damping, taps=["coll_G", "coll_h"])
chunks = np.array_split(np.arange(B), 64)
with mp.get_context("fork").Pool(16) as pool:
    parts = pool.map(T._numpy_oracle_chunk, [(name, c, q, tg, pt, dt, damping) for c in chunks])
v_ref = np.concatenate(parts)
err = np.abs(v - v_ref).max(axis=1) / np.maximum(1.0, np.abs(v_ref).max(axis=1))
bad = np.flatnonzero(err > 1e-6)
print("bad", len(bad), bad[:20], err[bad][:20])
np.savez("gpurun_out/convex_dbg.npz", q=q, tg=tg, v=v, v_ref=v_ref, st=st, coll_G=t["coll_G"], coll_h=t["coll_h"], bad=bad)
