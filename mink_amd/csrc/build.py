"""Build libminkhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

    python -m mink_amd.csrc.build            # or: python mink_amd/csrc/build.py

The shared library is written next to the Python package (mink_amd/libminkhip.so) so
that it travels with the repo snapshot; it is git-ignored (*.so).
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "libminkhip.so")
SOURCES = ["minkhip.hip"]
HEADERS = ["ik_kernel.h", "lie_dev.h", "collide_dev.h", "wave_ops.h", "mkh_types.h",
           os.path.join("..", "..", "include", "minkhip.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-o", LIB] + [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        print("[mink_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
