#!/usr/bin/env python
"""Compact timing lines of named workloads (bench.py's `other_configs` records), for same-box A/B runs:

    python tools/side_bench.py shadow_c4 ur5e_convex [--steps 20]
    MKH_LIB_TAG=x python tools/side_bench.py shadow_c4          # an experiment build (mink_amd/csrc/build.py MKH_BUILD_TAG)
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    import torch
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(bench.SIDE_CONFIGS)
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    dev = torch.device("cuda", 0)
    tag = os.environ.get("MKH_LIB_TAG", "") or "default"
    for n in names:
        batch = None
        if ":" in n:
            n, b = n.split(":"); batch = int(b)
        o = bench.measure_side_config(n, dev, steps=steps, batch=batch)
        print(f"[{tag}] {o['name']:12s} B={o['batch']:<7d} {o['kernel']:30s} {o['value'] / 1e6:8.2f} M/s  kernel {o['kernel_ms']:.4f} ms "
              f"(median {o['kernel_ms_median']:.4f})  hbm frac {100 * o['roofline']['frac']:.3f}%  failed {o['failed_instances']}", flush=True)


if __name__ == "__main__":
    main()
