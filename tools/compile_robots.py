"""Compile the benchmark robots' MJCF (from the reference's examples/, the only source of
robot models in the build image) into FlatModel JSON shipped with the package.

    python tools/compile_robots.py      # needs /root/reference

Generated data, not reference source: plain arrays of the mjModel fields listed in
mink_amd/flatmodel.py."""

import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from mink_amd.mjcf import load_mjcf  # noqa: E402

EX = "/root/reference/examples/"
ROBOTS = {
    "ur5e": EX + "universal_robots_ur5e/scene.xml",
    "g1": EX + "unitree_g1/scene.xml",
    "shadow_left": EX + "shadow_hand/scene_left.xml",
    "h1": EX + "unitree_h1/scene.xml",
    # the hand alone (no scene): what the reference's arm + hand examples attach to an arm (examples/arm_hand_iiwa_allegro.py:10-42);
    # mink_amd.compose.attach puts two of them on the G1's wrists for the `g1_hands` workload (75 dofs, 86 bodies)
    "allegro_left": EX + "wonik_allegro/left_hand.xml",
    # the reference's flagship collision example (examples/arm_aloha.py): arm / frame collision geoms are capsules FITTED to their
    # meshes at compile time (mink_amd/meshes.py), so the packaged arrays carry everything the `aloha_coll` workload reads
    "aloha": EX + "aloha/scene.xml",
}
out = os.path.join(REPO, "mink_amd", "robots")
os.makedirs(out, exist_ok=True)
for name, path in ROBOTS.items():
    m = load_mjcf(path)
    m.save(os.path.join(out, f"{name}.json"))
    print(name, "nq", m.nq, "nv", m.nv, "nbody", m.nbody)
