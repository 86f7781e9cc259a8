"""Batched `solve_ik` / `build_ik` with mink's signature (mink/solve_ik.py:43-105).

The Task/Limit objects are snapshotted into one device descriptor per call site
(cached on the Configuration by their constructor state); targets travel per call.
"""

from __future__ import annotations

import logging
from typing import NamedTuple, Optional, Sequence

import numpy as np

from . import _native as nat
from . import exceptions
from .configuration import Configuration
from .flatmodel import JNT_FREE

DEVICE_SOLVERS = ("mi355x", "hip", "quadprog")
PROBLEM_CACHE_SIZE = 16       # compiled descriptors kept per Configuration (least recently used are destroyed)


class Problem(NamedTuple):
    """The (P, q, G, h) of qpsolvers.Problem, possibly with a leading batch dimension."""
    P: np.ndarray
    q: np.ndarray
    G: Optional[np.ndarray]
    h: Optional[np.ndarray]


def _key(x):
    if isinstance(x, dict):
        return tuple((k, _key(v)) for k, v in sorted(x.items()))
    if isinstance(x, np.ndarray):
        return (x.shape, x.tobytes())
    if isinstance(x, (list, tuple)):
        return tuple(_key(v) for v in x)
    return x


def _limit_rows(configuration: Configuration, lim, dt: float):
    """(G, h) of a caller-defined limit for every instance: (B, m, nv), (B, m); m = 0 when inactive."""
    B, nv = configuration.batch_size, configuration.nv
    c = lim.compute_qp_inequalities(configuration, dt)
    if c.inactive:
        return np.zeros((B, 0, nv)), np.zeros((B, 0))
    G, h = np.asarray(c.G, dtype=np.float64), np.asarray(c.h, dtype=np.float64)
    m = h.shape[-1]
    if G.shape not in ((m, nv), (B, m, nv)) or h.shape not in ((m,), (B, m)):
        raise exceptions.LimitDefinitionError(
            f"{type(lim).__name__}.compute_qp_inequalities must return G ({m}, {nv}) or ({B}, {m}, {nv}) and h ({m},) "
            f"or ({B}, {m}); got {G.shape}, {h.shape}")
    return np.broadcast_to(G, (B, m, nv)), np.broadcast_to(h, (B, m))


def _fold_box_rows(G: np.ndarray, h: np.ndarray):
    """Split the rows G·Δq ≤ h of a caller-defined limit into per-dof BOX bounds and general half-spaces.

    A row whose only nonzero entry (over the whole batch) sits in one column k is g·Δq_k ≤ h: an upper bound h / g for
    g > 0, a lower bound for g < 0.  The reference stacks such rows into its dense G like any other
    (mink/solve_ik.py:25-40) and quadprog treats them as general constraints; on the device they join lo ≤ Δq ≤ hi and
    cost no tableau row — an acceleration-style limit [I; −I] (2·nv rows) would otherwise never fit next to
    64 − nv half-space rows.  Returns (lo, hi, G_rest, h_rest, any_single): (B, nv), (B, nv), (B, m', nv), (B, m'), and whether
    the limit HAS single-entry rows at all — the structural fact the handle's layout is keyed on (their current values may
    all be inactive, h = +inf: the same handle must serve the next call where they are not)."""
    B, m, nv = G.shape
    lo, hi = np.full((B, nv), -np.inf), np.full((B, nv), np.inf)
    if m == 0:
        return lo, hi, G, h, False
    pattern = (G != 0.0).any(axis=0)                                # (m, nv): columns a row ever touches
    single = pattern.sum(axis=1) == 1
    for r in np.flatnonzero(single):
        k = int(np.flatnonzero(pattern[r])[0])
        g, hr = G[:, r, k], h[:, r]
        with np.errstate(divide="ignore", invalid="ignore"):
            b = hr / g
        pos, neg, zero = g > 0.0, g < 0.0, g == 0.0
        hi[pos, k] = np.minimum(hi[pos, k], b[pos])
        lo[neg, k] = np.maximum(lo[neg, k], b[neg])
        bad = zero & (hr < 0.0)                                     # 0·Δq ≤ h < 0: infeasible, as the reference would find
        hi[bad, k], lo[bad, k] = -np.inf, np.inf
    keep = ~single
    return lo, hi, G[:, keep], h[:, keep], bool(single.any())


def _dense_inputs(configuration: Configuration, layout, dt: float):
    """Per-call arrays of the plugin route (mkh_solve_dense), None when the call site has no caller-defined rows."""
    if not layout["dense"] and not layout["dense_limits"]:
        return None
    if layout.get("dense_box") is not None:
        out_box = {"limit_lo": np.ascontiguousarray(layout["dense_box"][0]), "limit_hi": np.ascontiguousarray(layout["dense_box"][1])}
    else:
        out_box = {}
    out = {}
    if layout["dense"]:
        rows = [t._dense_rows(configuration) for t in layout["dense"]]
        out["task_e"] = np.ascontiguousarray(np.concatenate([e for e, _ in rows], axis=1))
        out["task_J"] = np.ascontiguousarray(np.concatenate([J for _, J in rows], axis=1))
    if layout["dense_limit_rows"]:
        rows = layout["dense_limit_data"]                             # evaluated by _compile at this call's dt
        out["limit_G"] = np.ascontiguousarray(np.concatenate([G for G, _ in rows], axis=1))
        out["limit_h"] = np.ascontiguousarray(np.concatenate([h for _, h in rows], axis=1))
    out.update(out_box)
    return out


def _compile(configuration: Configuration, tasks: Sequence, limits: Optional[Sequence], batch: int,
             dense_dt: float = 1.0):
    from .limits import ConfigurationLimit
    from .tasks import Task

    if limits is None:
        limits = configuration._default_limits                       # mink/solve_ik.py:28-29
        if limits is None:
            limits = configuration._default_limits = [ConfigurationLimit(configuration.model)]
    # Memo for control loops: the same task / limit objects with the same costs → the same handle and layout, without
    # rebuilding and hashing every descriptor (78 µs of a 185 µs G1 iteration; tools/bench_control_loop.py).
    fps = [x._fingerprint() for x in tasks] + [x._fingerprint() for x in limits]
    memo_key = None if any(f is None for f in fps) else (tuple(fps), len(tasks), batch)
    if memo_key is not None:
        hit = configuration._compile_memo.get(memo_key)
        if hit is not None and hit[0] in configuration._problems:
            prob = configuration._problems.pop(hit[0])
            configuration._problems[hit[0]] = prob                   # most recently used
            return prob, hit[1]
    groups = {"frame": [], "posture": [], "com": [], "cfg": [], "vel": [], "col": [], "dense": []}
    layout = {"frame": [], "posture": [], "com": [], "dense": [], "dense_limits": [], "dense_limit_rows": 0}
    for t in tasks:
        if t._is_dense():                                           # caller-defined Task subclass: dense rows
            kind, desc = Task._native_desc(t, configuration)
        else:
            kind, desc = t._native_desc(configuration)
        groups[kind].append(desc)
        layout[kind].append(t)
    for lim in limits:
        if lim._is_dense():                                         # caller-defined Limit subclass: dense rows
            layout["dense_limits"].append(lim)
            continue
        kind, desc = lim._native_desc()
        if kind in ("cfg", "vel") and len(desc["indices"]) == 0:
            continue                                                # inactive Constraint() (solve_ik.py:34)
        groups[kind].append(desc)
    if layout["dense_limits"]:
        # the row count of a plugin limit is only known from what it returns: evaluate once here (dt does not change
        # the shape), keep the rows for this call
        raw = [_limit_rows(configuration, lim, dense_dt) for lim in layout["dense_limits"]]
        folded = [_fold_box_rows(np.asarray(G), np.asarray(h)) for G, h in raw]
        layout["dense_limit_data"] = [(G, h) for _, _, G, h, _ in folded]
        layout["dense_limit_rows"] = sum(h.shape[-1] for _, h in layout["dense_limit_data"])
        lo = np.maximum.reduce([f[0] for f in folded]); hi = np.minimum.reduce([f[1] for f in folded])
        if any(f[4] for f in folded):          # (structural: a box limit whose rows are all inactive now keeps its handle)
            layout["dense_box"] = (lo, hi)
        # (more general rows than the 64 − nv a wavefront holds: the instances in which that many are active are solved again by
        #  the workgroup-per-problem kernel with every row, as the reference's np.vstack would — mink/solve_ik.py:25-40; its
        #  workspace holds 448 rows per instance)
        cap = 448
        if layout["dense_limit_rows"] > cap:
            names = ", ".join(type(lim).__name__ for lim in layout["dense_limits"])
            raise exceptions.LimitDefinitionError(
                f"caller-defined limits ({names}) contribute {layout['dense_limit_rows']} general rows G·Δq ≤ h (rows with a "
                f"single nonzero entry are folded into the per-dof box and do not count); at most {cap} half-space rows per "
                f"instance, shared with collision contacts")
    key = (_key(groups), batch, layout["dense_limit_rows"], layout.get("dense_box") is not None)
    cache = configuration._problems
    prob = cache.pop(key, None)
    if prob is None:
        kwargs = dict(frame_tasks=groups["frame"], posture_tasks=groups["posture"],
                      com_tasks=groups["com"], configuration_limits=groups["cfg"], velocity_limits=groups["vel"],
                      collision_limits=groups["col"], max_batch=batch, dense_tasks=groups["dense"],
                      dense_limit_rows=layout["dense_limit_rows"], dense_limit_box=layout.get("dense_box") is not None)
        if len(configuration.devices) > 1 and batch >= len(configuration.devices):
            from .distributed import ShardedProblem                  # one handle per listed device, rows split among them
            prob = ShardedProblem(configuration.model, configuration.devices, **kwargs)
        else:
            prob = nat.NativeProblem(configuration.native, **kwargs)
    cache[key] = prob                                               # (re)insert as most recently used
    # Costs, gains and lm_damping are part of the device descriptor, so a caller that retunes a cost every control
    # step compiles a new descriptor every step: bound the cache (LRU) and free the evicted device buffers.  A handle
    # that a caller up the stack is about to solve on is pinned: caller-defined tasks evaluate built-in ones
    # (Task._eval → _compile) between the outer _compile and its solve, and must not evict the outer handle.
    pinned = configuration._pinned_problems
    for old in [k for k in cache if k not in pinned][:max(0, len(cache) - PROBLEM_CACHE_SIZE)]:
        cache.pop(old).close()
    layout["cache_key"] = key
    if memo_key is not None:
        if len(configuration._compile_memo) >= 4 * PROBLEM_CACHE_SIZE:
            configuration._compile_memo.clear()
        configuration._compile_memo[memo_key] = (key, layout)
    return prob, layout


class _pin:
    """Keep a compiled handle out of LRU eviction while its caller still has to solve on it."""

    def __init__(self, configuration: Configuration, layout):
        self.pins, self.key = configuration._pinned_problems, layout["cache_key"]

    def __enter__(self):
        self.pins[self.key] = self.pins.get(self.key, 0) + 1

    def __exit__(self, *exc):
        n = self.pins[self.key] - 1
        if n:
            self.pins[self.key] = n
        else:
            del self.pins[self.key]


def _gather_targets(configuration: Configuration, layout):
    B = configuration.batch_size
    ft = pt = ct = None
    if layout["frame"]:
        ft = np.stack([t._native_target(configuration) for t in layout["frame"]], axis=1)
    if layout["posture"]:
        rows = [t._native_target(configuration) for t in layout["posture"]]
        if any(r.ndim == 2 for r in rows):
            pt = np.stack([np.broadcast_to(r, (B, r.shape[-1])) for r in rows], axis=1)
        else:
            pt = np.stack(rows, axis=0)
    if layout["com"]:
        rows = [t._native_target(configuration) for t in layout["com"]]
        if any(r.ndim == 2 for r in rows):
            ct = np.stack([np.broadcast_to(r, (B, 3)) for r in rows], axis=1)
        else:
            ct = np.stack(rows, axis=0)
    return ft, pt, ct


def build_ik(configuration: Configuration, tasks: Sequence, dt: float, damping: float = 1e-12,
             limits: Optional[Sequence] = None) -> Problem:
    """mink/solve_ik.py:43-65: the dense QP (P, q, G, h) — evaluated on the device, returned for
    inspection.  G/h stack the limits in list order exactly like the reference."""
    from .limits import ConfigurationLimit

    prob, layout = _compile(configuration, tasks, limits, configuration.batch_size, dt)
    ft, pt, ct = _gather_targets(configuration, layout)
    with _pin(configuration, layout):
        _, _, out = prob.solve(configuration.q_batch, ft, pt, ct, dt, damping, taps=["H", "c"], solve_qp=False,
                               dense=_dense_inputs(configuration, layout, dt))
    lims = [ConfigurationLimit(configuration.model)] if limits is None else limits
    G_list, h_list = [], []
    for lim in lims:
        c = lim.compute_qp_inequalities(configuration, dt)
        if not c.inactive:
            G_list.append(np.asarray(c.G)); h_list.append(np.asarray(c.h))
    un = configuration._unbatch
    if not G_list:
        return Problem(un(out["H"]), un(out["c"]), None, None)
    B = configuration.batch_size

    def bcast(a, nd):
        return a if (not configuration.batched or a.ndim == nd + 1) else np.broadcast_to(a, (B,) + a.shape)

    G = np.concatenate([bcast(g, 2) for g in G_list], axis=-2)
    h = np.concatenate([bcast(x, 1) for x in h_list], axis=-1)
    return Problem(un(out["H"]), un(out["c"]), G, h)


def solve_ik(configuration: Configuration, tasks: Sequence, dt: float, solver: str = "mi355x",
             damping: float = 1e-12, safety_break: bool = False, limits: Optional[Sequence] = None,
             return_status: bool = False, **kwargs) -> np.ndarray:
    """Velocity tangent to the batch of configurations (mink/solve_ik.py:68-105).

    `solver` is kept for signature compatibility; every value selects the device active-set
    solver (the reference forwards the string to qpsolvers).  Returns v of shape (nv,) or (B, nv).

    The reference forwards `**kwargs` to the QP solver; the one understood here is `warm_start=True` for closed loops
    (solve, integrate, solve again on the same batch): the active-set phase starts from where the previous solve of the
    same tasks / limits on this configuration ended (MKH_FLAG_WARM_START) — same optimum, fewer pivots once the loop has
    run for a few steps.  Other solver keywords are accepted and ignored.
    """
    warm_start = bool(kwargs.pop("warm_start", False))
    del kwargs
    prob, layout = _compile(configuration, tasks, limits, configuration.batch_size, dt)
    ft, pt, ct = _gather_targets(configuration, layout)
    with _pin(configuration, layout):
        v, status = prob.solve(configuration.q_batch, ft, pt, ct, dt, damping,
                               dense=_dense_inputs(configuration, layout, dt), warm_start=warm_start)
    if (status & nat.ST_OUTSIDE_LIMITS).any():
        configuration.check_limits(safety_break=safety_break)      # raises / warns like the reference
    bad = np.nonzero(status & ~nat.ST_OUTSIDE_LIMITS)[0]
    if len(bad):
        s = int(status[bad[0]])
        why = ("constraints are inconsistent, no solution" if s & nat.ST_INFEASIBLE else
               "matrix P is not positive definite" if s & nat.ST_NOT_PD else
               "active-set iteration limit reached" if s & nat.ST_ITER_LIMIT else
               "more simultaneous contacts than tableau rows")
        raise exceptions.SolverError(f"QP failed for {len(bad)} of {len(status)} instances "
                                     f"(first: index {int(bad[0])}, status {s}): {why}")
    v = configuration._unbatch(v)
    if return_status:
        return v, configuration._unbatch(status)
    return v


def solve_ik_steps(configuration: Configuration, tasks: Sequence, dt: float, n_steps: int,
                   solver: str = "mi355x", damping: float = 1e-12, safety_break: bool = False,
                   limits: Optional[Sequence] = None, update: bool = True,
                   pos_threshold: Optional[float] = None, ori_threshold: Optional[float] = None):
    """`n_steps` iterations of  v = solve_ik(...); configuration.integrate_inplace(v, dt)  fused in one
    kernel launch (the loop mink's callers write themselves, e.g. examples/arm_ur5e_actuators.py:88-97).

    Returns (q_final, v_last); with `update` the configuration is advanced in place.

    With `pos_threshold` / `ori_threshold` the loop is the callers' real one — it breaks, per instance, as soon as
    every frame task's error is within the thresholds after the integration (arm_ur5e_actuators.py:93-97), `n_steps`
    is max_iters — and the return value is (q_final, v_last, iters, converged)."""
    until = None
    if pos_threshold is not None or ori_threshold is not None:
        until = (float(pos_threshold if pos_threshold is not None else np.inf),
                 float(ori_threshold if ori_threshold is not None else np.inf))
    prob, layout = _compile(configuration, tasks, limits, configuration.batch_size, dt)
    if layout["dense"] or layout["dense_limits"]:
        raise exceptions.TaskDefinitionError(
            "solve_ik_steps fuses the outer loop on the device; caller-defined Task / Limit subclasses are evaluated on "
            "the host at every step: call solve_ik + integrate_inplace in a loop instead")
    ft, pt, ct = _gather_targets(configuration, layout)
    res = prob.solve(configuration.q_batch, ft, pt, ct, dt, damping, n_steps=int(n_steps), until=until)
    q, v, status = res[:3]
    if (status & nat.ST_OUTSIDE_LIMITS).any():
        # The bit is the OR over the fused steps (the reference loop checks every iteration, solve_ik.py:97): the
        # start configuration first, then — a violation that appeared at step k > 0 — the last one.  An instance
        # that left and re-entered its limits in between only warns.
        # With safety_break the two check_limits calls raise NotWithinConfigurationLimits like the reference; without
        # it ONE warning is logged (a control loop calls this every tick: no fresh Configuration, no warning per joint).
        if safety_break:
            configuration.check_limits(safety_break=True)
            Configuration(configuration.model, q, device=configuration.device).check_limits(safety_break=True)
        logging.warning("solve_ik_steps: %d instance(s) were outside their configuration limits at some fused step",
                        int(((status & nat.ST_OUTSIDE_LIMITS) != 0).sum()))
    bad = np.nonzero(status & ~nat.ST_OUTSIDE_LIMITS)[0]
    if len(bad):
        raise exceptions.SolverError(f"QP failed for {len(bad)} of {len(status)} instances "
                                     f"(first: index {int(bad[0])}, status {int(status[bad[0]])})")
    if update:
        configuration.update(q if configuration.batched else q[0])
    if until is not None:
        return (configuration._unbatch(q), configuration._unbatch(v), configuration._unbatch(res[3]),
                configuration._unbatch(res[4].astype(bool)))
    return configuration._unbatch(q), configuration._unbatch(v)
