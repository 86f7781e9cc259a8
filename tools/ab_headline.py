#!/usr/bin/env python
"""Same-box A/B of library builds on the headline workload: alternates the builds (MKH_LIB_TAG values, '' = product) in
separate processes, several rounds, and prints the kernel ms of each.    python tools/ab_headline.py "" np [--config g1_c3]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "g1_c3"
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
code = ("import sys, json; sys.path.insert(0, %r); import bench, torch; "
        "o = bench.measure_side_config(%r, torch.device('cuda', 0), steps=30, warmup=5); "
        "print('%%s %%.4f %%.4f' %% (o['kernel'], o['kernel_ms'], o['kernel_ms_median']))") % (REPO, cfg)
res = {t: [] for t in args}
for r in range(rounds):
    for t in args:
        env = dict(os.environ, MKH_LIB_TAG=t)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        res[t].append(out)
for t in args:
    print("[%s]" % (t or "product"), " | ".join(res[t]))
