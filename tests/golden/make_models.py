"""FlatModels of EVERY example scene of the reference (examples/*/scene*.xml), compiled by the MJCF reader in the
build container and committed, so that the GPU box — which has no /root/reference — can solve them
(tests/test_gpu_all_robots.py).  Build container only:

    python tests/golden/make_models.py
"""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from mink_amd.mjcf import load_mjcf  # noqa: E402


def main():
    out = os.path.join(HERE, "models", "all")
    os.makedirs(out, exist_ok=True)
    for p in sorted(glob.glob("/root/reference/examples/*/scene*.xml")):
        name = p.split("/")[-2] + "__" + os.path.basename(p)[:-4]
        m = load_mjcf(p)
        m.save(os.path.join(out, name + ".json"))
        print(name, "nv", m.nv, "nbody", m.nbody)


if __name__ == "__main__":
    main()
