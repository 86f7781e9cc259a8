"""Random kinematic trees as MJCF text (shared by the CPU oracle-vs-oracle test and the GPU parity test)."""

import numpy as np

from oracle import lie as olie


def random_mjcf(rng, nbody, free_root, no_ball=False):
    """A random tree of `nbody` moving bodies as MJCF text (+ the list of site names).  no_ball: hinge / slide joints
    only (what the row- and lane-per-problem kernels of small arms take)."""
    parent = [-1] + [int(rng.integers(0, i)) for i in range(1, nbody)]
    if rng.uniform() < 0.5:                       # a long chain somewhere: depth matters for pointer jumping
        for i in range(1, nbody // 2):
            parent[i] = i - 1
    children = {i: [] for i in range(-1, nbody)}
    for i, p in enumerate(parent):
        children[p].append(i)
    sites = []

    def fmt(v):
        return " ".join(f"{x:.6f}" for x in v)

    def body(i, depth):
        pad = "  " * (depth + 2)
        pos = rng.normal(scale=0.15, size=3)
        quat = olie.so3_exp(rng.normal(scale=0.6, size=3))
        out = [f'{pad}<body name="b{i}" pos="{fmt(pos)}" quat="{fmt(quat)}">']
        out.append(f'{pad}  <inertial pos="{fmt(rng.normal(scale=0.03, size=3))}" mass="{rng.uniform(0.2, 2.0):.4f}" diaginertia="1 1 1"/>')
        if i == 0 and free_root:
            out.append(f'{pad}  <freejoint name="root"/>')
        else:
            kinds = rng.choice(["hinge", "hinge", "hinge", "slide", "ball", "fixed", "two"], p=[.3, .2, .1, .12, .1, .08, .1])
            if no_ball and kinds == "ball":
                kinds = "slide"
            if kinds == "ball":
                out.append(f'{pad}  <joint name="j{i}" type="ball" pos="{fmt(rng.normal(scale=0.02, size=3))}"/>')
            elif kinds == "fixed" and i > 0:
                pass
            else:
                n = 2 if kinds == "two" else 1
                for k in range(n):
                    typ = "slide" if kinds == "slide" else "hinge"
                    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
                    lim = ""
                    if rng.uniform() < 0.7:
                        lo = -rng.uniform(0.3, 1.5); hi = rng.uniform(0.3, 1.5)
                        if typ == "slide":
                            lo, hi = 0.3 * lo, 0.3 * hi
                        lim = f' range="{lo:.4f} {hi:.4f}"'
                    out.append(f'{pad}  <joint name="j{i}_{k}" type="{typ}" axis="{fmt(ax)}" pos="{fmt(rng.normal(scale=0.02, size=3))}"{lim}/>')
        if rng.uniform() < 0.6:
            sites.append(f"s{i}")
            out.append(f'{pad}  <site name="s{i}" pos="{fmt(rng.normal(scale=0.08, size=3))}" quat="{fmt(olie.so3_exp(rng.normal(size=3)))}"/>')
        for c in children[i]:
            out += body(c, depth + 1)
        out.append(f"{pad}</body>")
        return out

    lines = ['<mujoco>', '  <compiler angle="radian" autolimits="true"/>', '  <worldbody>']
    for r in children[-1]:
        lines += body(r, 0)
    lines += ['  </worldbody>', '</mujoco>']
    return "\n".join(lines), sites


def rand_q(m, rng):
    q = np.array(m.qpos0)
    for j in range(m.njnt):
        a, t = int(m.jnt_qposadr[j]), int(m.jnt_type[j])
        if t == 0:
            q[a:a + 3] += rng.normal(scale=0.2, size=3)
            q[a + 3:a + 7] = olie.so3_exp(rng.normal(scale=0.5, size=3))
        elif t == 1:
            q[a:a + 4] = olie.so3_exp(rng.normal(scale=0.7, size=3))
        else:
            lo, hi = m.jnt_range[j] if m.jnt_limited[j] else (-1.0, 1.0)
            q[a] = rng.uniform(lo + 0.02 * (hi - lo), hi - 0.02 * (hi - lo))
    return q




def chain_mjcf(ndof, seed=0, sites_every=25):
    """A serial chain of `ndof` one-dof bodies (hinge, every seventh a slide; all limited) — one body per dof, so
    ndof = 100 is 101 bodies with the world: past both one-wavefront limits.  Sites every `sites_every` links and at the
    tip."""
    rng = np.random.default_rng(seed)

    def fmt(v):
        return " ".join(f"{x:.6f}" for x in v)

    sites = []
    lines = ['<mujoco>', '  <compiler angle="radian" autolimits="true"/>', '  <worldbody>']
    depth = 0
    for i in range(ndof):
        pad = "  " * (depth + 2)
        pos = np.array([0.0, 0.0, 0.04]) + rng.normal(scale=0.01, size=3)
        quat = olie.so3_exp(rng.normal(scale=0.3, size=3))
        lines.append(f'{pad}<body name="b{i}" pos="{fmt(pos)}" quat="{fmt(quat)}">')
        lines.append(f'{pad}  <inertial pos="0 0 0.02" mass="{rng.uniform(0.1, 0.5):.4f}" diaginertia="1 1 1"/>')
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        if i % 7 == 6:
            lines.append(f'{pad}  <joint name="j{i}" type="slide" axis="{fmt(ax)}" range="-0.02 0.02"/>')
        else:
            lines.append(f'{pad}  <joint name="j{i}" type="hinge" axis="{fmt(ax)}" range="-0.6 0.6"/>')
        if (i + 1) % sites_every == 0 or i == ndof - 1:
            sites.append(f"s{i}")
            lines.append(f'{pad}  <site name="s{i}" pos="0.01 0 0.03"/>')
        depth += 1
    for i in reversed(range(ndof)):
        lines.append("  " * (i + 2) + "</body>")
    lines += ['  </worldbody>', '</mujoco>']
    return "\n".join(lines), sites
