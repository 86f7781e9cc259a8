"""GPU parity against the fixtures recorded from the real mink (tests/golden/ik_*.npz):
every intermediate the reference exposes (frame poses, task errors/Jacobians, H, c,
box/half-space limits) and the final velocity, through the C ABI (host pointers)."""

import os

import numpy as np
import pytest

import native_configs as nc
import oracle_configs as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from mink_amd import _native
    assert _native.lib().mkh_device_count() >= 1
    return _native


def _golden(name):
    return np.load(os.path.join(oc.GOLDEN, f"ik_{name}.npz"))


def _reorder_rows(name, m):
    """golden task rows follow the mink task list order; native taps are frame, posture, com."""
    nv = m.nv
    if name == "ur5e_c2" or name == "g1_c3":
        return None
    if name == "g1_full":   # [pelvis(6), posture(nv), com(3), feet(12), hands(12)]
        idx = list(range(0, 6)) + list(range(6 + nv + 3, 6 + nv + 3 + 24)) + list(range(6, 6 + nv)) + \
            list(range(6 + nv, 6 + nv + 3))
        return np.array(idx)
    if name == "shadow_c4":  # [posture(nv), fingers(30)]
        return np.array(list(range(nv, nv + 30)) + list(range(nv)))
    raise KeyError(name)


@pytest.mark.parametrize("name", ["ur5e_c2", "g1_c3", "g1_full", "shadow_c4"])
def test_intermediates_and_velocity(nat, name):
    d = _golden(name)
    m = oc.model(nc.ROBOT_OF[name])
    nm = nat.NativeModel(m)
    B = len(d["q"])
    prob, dt, damping = nc.build(name, nm, B)
    assert dt == float(d["dt"]) and damping == float(d["damping"])
    taps = ["frame_pose", "task_e", "task_J", "H", "c", "box_lo", "box_hi", "qp_iters"]
    if prob.n_pairs:
        taps += ["coll_G", "coll_h"]
    com = d["com_target"][:, None, :] if "com_target" in d.files else None
    v, st, t = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], com, dt, damping, taps=taps)
    rows = _reorder_rows(name, m)
    e_ref = d["task_e"] if rows is None else d["task_e"][:, rows]
    np.testing.assert_allclose(t["task_e"], e_ref, rtol=0, atol=1e-12)
    J_ref = d["task_J"] if rows is None else d["task_J"][:, rows]
    np.testing.assert_allclose(t["task_J"][:len(J_ref)], J_ref, rtol=0, atol=1e-9)
    # main stream vs small-angle sub-stream (every 8th sample has |δ| ~ 1e-4: the reference's own
    # jlog carries ≈5e-17/θ² cancellation noise there — SURVEY §7 hard part 3 — so it gets 1e-6)
    main = np.ones(B, bool); main[7::8] = False
    scale = np.abs(d["H"]).max(axis=(1, 2), keepdims=True)
    dH = np.abs(t["H"] - d["H"]) / scale
    cs = np.abs(d["c"]).max(axis=1, keepdims=True)
    dc = np.abs(t["c"] - d["c"]) / cs
    print(name, "H err main/small", dH[main].max(), dH[~main].max(), "c err", dc[main].max(), dc[~main].max())
    assert dH[main].max() < 1e-11 and dc[main].max() < 1e-11
    assert dH[~main].max() < 1e-6 and dc[~main].max() < 1e-6
    # box limits vs the reference's stacked rows G=[P;-P] (configuration, then velocity)
    h = d["h"]
    idx = np.array([int(m.jnt_dofadr[j]) for j in range(m.njnt) if m.jnt_type[j] != 0 and m.jnt_limited[j]])
    n = len(idx)
    hi = h[:, :n].copy(); lo = -h[:, n:2 * n]
    if name in ("ur5e_c2", "g1_c3", "g1_full"):
        hi = np.minimum(hi, h[:, 2 * n:3 * n]); lo = np.maximum(lo, -h[:, 3 * n:4 * n])
    np.testing.assert_allclose(t["box_hi"][:, idx], hi, rtol=0, atol=1e-14)
    np.testing.assert_allclose(t["box_lo"][:, idx], lo, rtol=0, atol=1e-14)
    if prob.n_pairs:
        hc = h[:, 2 * n:]
        fin = np.isfinite(hc)
        assert (np.isfinite(t["coll_h"]) == fin).all()
        np.testing.assert_allclose(t["coll_h"][fin], hc[fin], rtol=0, atol=1e-9)
        G_ref = d["G"][:, 2 * n:]
        np.testing.assert_allclose(t["coll_G"][:len(G_ref)], G_ref, rtol=0, atol=1e-10)
    assert (st & ~1 == 0).all(), st
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    err = np.abs(v - d["v"]) / vs
    print(name, "max rel v err", err.max(), "qp iters", t["qp_iters"].tolist())
    # stated fp64 tolerance: 1e-8 relative on the main stream; the small-angle
    # sub-stream (every 8th sample, δ ~ 1e-4) is allowed 1e-5 (reference jlog noise ~5e-17/θ²)
    assert err[main].max() < 1e-8
    assert err[~main].max() < 1e-5
    # the production call (no taps) runs the lean kernel variant — for g1_c3 with the low-rank QP start
    v2, st2 = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], com, dt, damping)
    err2 = np.abs(v2 - d["v"]) / vs
    print(name, "production kernel", prob.last_kernel(), "max rel v err", err2.max())
    if name == "g1_c3":
        assert prob.last_kernel().removesuffix("o").endswith("_32_r44_w3"), prob.last_kernel()   # low-rank start, 44 dof rows, 3 waves per SIMD
    assert (st2 == st).all()
    assert err2[main].max() < 1e-8 and err2[~main].max() < 1e-5


def _box_from_rows(G, h, nv):
    """lo ≤ Δq ≤ hi implied by the ±unit rows of a reference (G, h) (configuration / velocity limit rows)."""
    lo, hi = np.full(nv, -np.inf), np.full(nv, np.inf)
    used = np.zeros(len(h), bool)
    for r in range(len(h)):
        nz = np.flatnonzero(G[r])
        if len(nz) == 1 and abs(G[r, nz[0]]) == 1.0:
            k = nz[0]
            used[r] = True
            if G[r, k] > 0:
                hi[k] = min(hi[k], h[r])
            else:
                lo[k] = max(lo[k], -h[r])
    return lo, hi, used


@pytest.mark.parametrize("name", ["g1_ext", "ur5e_coll", "ballslide", "balllimit"])
def test_round2_fixtures(nat, name):
    """Fixtures recorded from the real mink by tests/golden/make_golden_ext.py: RelativeFrameTask (moving roots),
    DampingTask, body/geom frames, PER-INSTANCE posture and CoM targets (MKH_FLAG_POSTURE_BATCHED / _COM_BATCHED),
    the arm_ur5e.py collision set-up (capsule vs plane / box, half-spaces binding), ball + slide joints."""
    d = _golden(name)
    m = oc.model(nc.ROBOT_OF[name])
    nm = nat.NativeModel(m)
    B = len(d["q"])
    prob, (ft, pt, ct), rows, dt, damping = nc.build_ext(name, nm, d, B)
    assert dt == float(d["dt"]) and damping == float(d["damping"])
    taps = ["task_e", "task_J", "H", "c", "box_lo", "box_hi"] + (["coll_G", "coll_h"] if prob.n_pairs else [])
    v, st, t = prob.solve(d["q"], ft, pt, ct, dt, damping, taps=taps)
    e_ref = d["task_e"] if rows is None else d["task_e"][:, rows]
    np.testing.assert_allclose(t["task_e"], e_ref, rtol=0, atol=1e-12)
    J_ref = d["task_J"] if rows is None else d["task_J"][:, rows]
    np.testing.assert_allclose(t["task_J"][:len(J_ref)], J_ref, rtol=0, atol=1e-9)
    main = np.ones(B, bool)
    if name not in ("ur5e_coll", "balllimit"):
        main[7::8] = False                                    # small-angle sub-stream (δ ~ 1e-4)
    dH = np.abs(t["H"] - d["H"]) / np.abs(d["H"]).max(axis=(1, 2), keepdims=True)
    dc = np.abs(t["c"] - d["c"]) / np.abs(d["c"]).max(axis=1, keepdims=True)
    print(name, "H err main/small", dH[main].max(), dH[~main].max(initial=0), "c err", dc[main].max(), dc[~main].max(initial=0))
    assert dH[main].max() < 1e-11 and dc[main].max() < 1e-11
    assert dH[~main].max(initial=0) < 1e-6 and dc[~main].max(initial=0) < 1e-6
    nG = len(d["G"])
    for i in range(nG):
        lo, hi, used = _box_from_rows(d["G"][i], d["h"][i], m.nv)
        np.testing.assert_allclose(t["box_lo"][i], lo, rtol=0, atol=1e-13)
        np.testing.assert_allclose(t["box_hi"][i], hi, rtol=0, atol=1e-13)
        if prob.n_pairs:
            hc, Gc = d["h"][i][~used], d["G"][i][~used]
            fin = np.isfinite(hc)
            assert (np.isfinite(t["coll_h"][i]) == fin).all()
            np.testing.assert_allclose(t["coll_h"][i][fin], hc[fin], rtol=0, atol=1e-9)
            np.testing.assert_allclose(t["coll_G"][i], Gc, rtol=0, atol=1e-10)
    assert (st & ~1 == 0).all(), st
    if name == "balllimit":
        # the reference's check_limits compares the limited ball joint's quaternion w with its range (a warning under
        # safety_break=False): the device sets the same per-instance bit
        m_ = oc.model("balllimit")
        j = m_.name2id("joint", "ball2")
        qw = d["q"][:, m_.jnt_qposadr[j]]
        expect = (qw < m_.jnt_range[j, 0] - 1e-6) | (qw > m_.jnt_range[j, 1] + 1e-6)
        assert ((st & 1).astype(bool) == expect).all() and expect.any()
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    err = np.abs(v - d["v"]) / vs
    print(name, "max rel v err main/small", err[main].max(), err[~main].max(initial=0))
    assert err[main].max() < 1e-8 and err[~main].max(initial=0) < 1e-5
    v2, st2 = prob.solve(d["q"], ft, pt, ct, dt, damping)       # the production call (no taps)
    err2 = np.abs(v2 - d["v"]) / vs
    print(name, "production kernel", prob.last_kernel(), "max rel v err", err2.max())
    assert (st2 == st).all() and err2[main].max() < 1e-8 and err2[~main].max(initial=0) < 1e-5
    if name == "ur5e_coll":
        # the fixture is only worth its name if half-spaces really bind
        n = 6
        hc = d["h"][:, 2 * n:2 * n + 2]
        Gx = np.einsum("bpj,bj->bp", d["G"][:, 2 * n:2 * n + 2], v * dt)
        assert ((np.abs(Gx - hc) < 1e-9) & np.isfinite(hc)).sum() >= 2


def test_ur5e_c1_trajectory_replay(nat):
    """BASELINE configs[0]: the real mink's 20-step closed loop on one UR5e instance, `limits=None`
    (⇒ a fresh ConfigurationLimit, /root/reference/mink/solve_ik.py:28-29; the convergence test of the reference,
    tests/test_solve_ik.py:95-148).  Every recorded (q_k → v_k) through mkh_solve at B = 1, then the whole loop through
    mkh_solve_steps(20) → q_final."""
    d = _golden("ur5e_c1")
    m = oc.model("ur5e")
    nm = nat.NativeModel(m)
    prob, dt, damping = nc.build("ur5e_c1", nm, 1)
    assert dt == float(d["dt"]) and damping == float(d["damping"])
    tgt = d["frame_target"][None, None, :]
    post = d["posture_target"][None, :]
    worst = 0.0
    for k in range(len(d["q"])):
        v, st = prob.solve(d["q"][k][None, :], tgt, post, None, dt, damping)
        assert st[0] == 0
        worst = max(worst, float(np.abs(v[0] - d["v"][k]).max() / max(1.0, np.abs(d["v"][k]).max())))
    assert worst < 1e-8, worst
    qf, v_last, st = prob.solve(d["q"][0][None, :], tgt, post, None, dt, damping, n_steps=len(d["q"]))
    assert st[0] == 0
    np.testing.assert_allclose(qf[0], d["q_final"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v_last[0], d["v"][-1], rtol=0, atol=1e-8 * max(1.0, np.abs(d["v"][-1]).max()))
    # the same trajectory with the same B through the lane-per-problem kernel family (pinned by flag)
    qf2, _, st2 = prob.solve(d["q"][0][None, :], tgt, post, None, dt, damping, n_steps=len(d["q"]), lane_kernel=True)
    assert st2[0] == 0
    np.testing.assert_allclose(qf2[0], d["q_final"], rtol=0, atol=1e-9)
    print("ur5e_c1 replay: max rel |v - mink| over 20 steps", worst, "| q_final err", float(np.abs(qf[0] - d["q_final"]).max()))


def test_production_kernel_against_2048_real_mink_instances(nat):
    """tests/golden/ik_g1_c3_big.npz (make_golden_big.py: the REAL mink on 2 048 instances of config 3) against the
    production kernel — low-rank start + cold-start refinement, 3 waves per SIMD — at a size the small fixtures never reach: several
    problems per persistent wavefront and the ticket-counter tail.  Then the other
    routes to the same answer: direct QP start, 2-waves register map, device pointers."""
    d = _golden("g1_c3_big")
    m = oc.model("g1")
    nm = nat.NativeModel(m)
    B = len(d["q"])
    prob, dt, damping = nc.build("g1_c3", nm, B)
    assert dt == float(d["dt"]) and damping == float(d["damping"])
    main = np.ones(B, bool); main[7::8] = False
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))

    def check(v, st, what):
        assert (st & ~1 == 0).all(), (what, np.flatnonzero(st & ~1)[:10])
        err = np.abs(v - d["v"]) / vs
        print(f"g1_c3_big {what} ({prob.last_kernel()}): max rel v err main {err[main].max():.2e} small-angle {err[~main].max():.2e}")
        assert err[main].max() < 1e-8 and err[~main].max() < 1e-5

    v, st = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping)
    assert prob.last_kernel().removesuffix("o") == "ik_solve_kernel_44_32_r44_w3", prob.last_kernel()
    check(v, st, "production")
    v2, st2 = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping, direct_qp=True)
    check(v2, st2, "direct start")
    v3, st3 = prob.solve(d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping, two_waves=True)
    check(v3, st3, "2-waves map")
    np.testing.assert_array_equal(v3, v)          # the two register maps run the same arithmetic
    # 2 048 problems are one per resident wavefront: tile the fixture — same inputs, so the same real-mink answers — through
    # the other two launch shapes of minkhip.hip::launch (round 5): × 8 = 16 384 instances, 5.3 rounds of the resident
    # wavefronts: one problem per WORKGROUP (the dispatcher deals them); × 40 = 81 920 instances, 26.7 rounds: persistent
    # wavefronts, a static share each and the tail of the batch through the ticket counter (round 6: on the two-waves build — the
    # three-waves build has a twin compiled for one problem per workgroup, `_w3o`, and runs it from 3.5 rounds on without an upper end)
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    to = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    for R, persistent in ((8, False), (40, True)):
        probR, _, _ = nc.build("g1_c3", nm, R * B)
        # (device pointers: a host-pointer call of this size would be cut into chunks, each a launch of its own size)
        vR, stR = probR.solve(to(np.tile(d["q"], (R, 1))), to(np.tile(d["frame_targets"], (R, 1, 1))), to(d["posture_target"][None, :]),
                              None, dt, damping, two_waves=persistent)
        torch.cuda.synchronize()
        vR, stR = vR.cpu().numpy(), stR.cpu().numpy()
        assert probR.last_kernel() == ("ik_solve_kernel_44_32_r44" if persistent else "ik_solve_kernel_44_32_r44_w3o"), probR.last_kernel()
        info = probR.launch_info(R * B)
        if persistent:
            assert R * B >= 8 * info["grid"], info             # ⇒ dynamic tail (minkhip.hip::launch, kMinRoundsForTickets)
        else:
            assert info["grid"] == R * B, info
        for r in range(R):
            np.testing.assert_array_equal(vR[r * B:(r + 1) * B], v)
            assert (stR[r * B:(r + 1) * B] == st).all()


def test_production_collision_path_against_2048_real_mink_instances(nat):
    """tests/golden/ik_shadow_c4_big.npz (make_golden_big2.py: the REAL mink on 2 048 instances of config 4, half of them
    pulled towards `grasp hard`: 17 contacts in range on average, up to 33) against the PLAIN call — the tight-rows launch
    `48_72` with 24 rows for the 40 pairs plus the full-row redo launch for the instances it flags, whose row selection cannot be
    tapped — and against the full-row build by flag.  The contact bounds h of all 40 pairs ride along (taps, parity build)."""
    d = _golden("shadow_c4_big")
    m = oc.model("shadow_left")
    nm = nat.NativeModel(m)
    B = len(d["q"])
    prob, dt, damping = nc.build("shadow_c4", nm, B)
    assert dt == float(d["dt"]) and damping == float(d["damping"])
    main = np.ones(B, bool); main[7::8] = False
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    fin = np.isfinite(d["coll_h"])
    assert fin.sum(axis=1).max() > 24            # more contacts in range than tight rows somewhere: the redo launch has work

    def check(v, st, what):
        assert (st & ~1 == 0).all(), (what, np.unique(st, return_counts=True))
        err = np.abs(v - d["v"]) / vs
        print(f"shadow_c4_big {what} ({prob.last_kernel()}): max rel v err main {err[main].max():.2e} small-angle {err[~main].max():.2e}")
        assert err[main].max() < 1e-8 and err[~main].max() < 1e-5

    args = (d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping)
    v, st = prob.solve(*args)
    assert prob.last_kernel().removesuffix("+wide") == "ik_solve_kernel_48_72+redo_64", prob.last_kernel()
    check(v, st, "tight rows + redo")
    v2, st2 = prob.solve(*args, full_rows=True)
    assert prob.last_kernel().removesuffix("+wide") == "ik_solve_kernel_64_72", prob.last_kernel()
    check(v2, st2, "full rows")
    _, _, t = prob.solve(*args, taps=["coll_h"])
    assert (np.isfinite(t["coll_h"]) == fin).all()
    np.testing.assert_allclose(t["coll_h"][fin], d["coll_h"][fin], rtol=0, atol=1e-9 * max(1.0, np.abs(d["coll_h"][fin]).max()))
    # tiled ×8 = the BASELINE batch: persistent wavefronts run several rounds, the redo launch scans 16 384 statuses
    R = 8
    probR, _, _ = nc.build("shadow_c4", nm, R * B)
    vR, stR = probR.solve(np.tile(d["q"], (R, 1)), np.tile(d["frame_targets"], (R, 1, 1)), d["posture_target"][None, :], None, dt, damping)
    assert probR.last_kernel().removesuffix("+wide") == "ik_solve_kernel_48_72+redo_64"
    for r in range(R):
        np.testing.assert_array_equal(vR[r * B:(r + 1) * B], v)
        assert (stR[r * B:(r + 1) * B] == st).all()


def test_small_arm_kernels_against_4096_real_mink_instances(nat):
    """tests/golden/ik_ur5e_c2_big.npz (make_golden_big2.py: the REAL mink on 4 096 instances of config 2, the batch the
    metric is quoted on) on the row kernel (the default at this size) and, by flag, the lane and wavefront kernels."""
    d = _golden("ur5e_c2_big")
    m = oc.model("ur5e")
    nm = nat.NativeModel(m)
    B = len(d["q"])
    prob, dt, damping = nc.build("ur5e_c2", nm, B)
    assert dt == float(d["dt"]) and damping == float(d["damping"])
    main = np.ones(B, bool); main[7::8] = False
    vs = np.maximum(1.0, np.abs(d["v"]).max(axis=1, keepdims=True))
    args = (d["q"], d["frame_targets"], d["posture_target"][None, :], None, dt, damping)
    for flags, kernel in (({}, "ik_quad_kernel"), ({"lane_kernel": True}, "ik_lane_kernel"), ({"wave_kernel": True}, "ik_solve_kernel")):
        v, st = prob.solve(*args, **flags)
        assert prob.last_kernel().startswith(kernel), prob.last_kernel()
        assert (st & ~1 == 0).all(), (kernel, np.unique(st, return_counts=True))
        err = np.abs(v - d["v"]) / vs
        print(f"ur5e_c2_big ({prob.last_kernel()}): max rel v err main {err[main].max():.2e} small-angle {err[~main].max():.2e}")
        assert err[main].max() < 1e-8 and err[~main].max() < 1e-5


@pytest.mark.parametrize("name", ["aloha_coll", "shadow_tips"])
def test_mesh_dependent_collision_setups_against_real_mink(nat, name):
    """tests/golden/make_golden_mesh.py: the REAL mink on the two set-ups whose collision geoms come from mesh assets —
    examples/arm_aloha.py:76-121 as written (1 104 pairs of mesh-fitted capsules, filtered by mink's own constructor out of
    whole subtrees) and the Shadow hand's `*_3` fingertips + forearm MESH geom.  Same pair list in the same order, h of every
    pair, the rows of G, v."""
    import mink_amd as mink
    from mink_amd.flatmodel import FlatModel
    d = _golden(name)
    scene = {"aloha_coll": "aloha__scene", "shadow_tips": "shadow_hand__scene_left"}[name]
    m = FlatModel.load(os.path.join(oc.GOLDEN, "models", "all", scene + ".json"))
    B = len(d["q"])
    cfg = mink.Configuration(m, d["q"])
    if name == "aloha_coll":
        sites, post_cost = ["left/gripper", "right/gripper"], 1e-4
        lw = mink.get_subtree_geom_ids(m, m.body("left/wrist_link").id); rw = mink.get_subtree_geom_ids(m, m.body("right/wrist_link").id)
        lg = mink.get_subtree_geom_ids(m, m.body("left/upper_arm_link").id); rg = mink.get_subtree_geom_ids(m, m.body("right/upper_arm_link").id)
        fg = mink.get_body_geom_ids(m, m.body("metal_frame").id)
        col = mink.CollisionAvoidanceLimit(m, [(lw, rw), (lg + rg, fg + ["table"])], minimum_distance_from_collisions=0.05,
                                           collision_detection_distance=0.1)
        vel = mink.VelocityLimit(m, {f"{p}/{n}": np.pi for p in ("left", "right")
                                     for n in ("waist", "shoulder", "elbow", "forearm_roll", "wrist_angle", "wrist_rotate")})
        lims = [mink.ConfigurationLimit(m), vel, col]
        fts = [mink.FrameTask(s, "site", 1.0, 1.0, lm_damping=1.0) for s in sites]
        tasks_order = lambda fts, post: fts + [post]
    else:
        sites, post_cost = list(oc.SHADOW_FINGERS), 1e-2
        fore = [g for g in range(m.ngeom) if m.geom_type[g] == 7 and m.geom_dataid[g] >= 0]
        tips, mids = [f"{f}_3" for f in sites], [f"{f}_2" for f in sites]
        col = mink.CollisionAvoidanceLimit(m, [(tips, tips), (tips, mids), (tips, fore)], minimum_distance_from_collisions=0.004,
                                           collision_detection_distance=0.06)
        lims = [mink.ConfigurationLimit(m), col]
        fts = [mink.FrameTask(s, "site", 1.0, 0.0, lm_damping=1.0) for s in sites]
        tasks_order = lambda fts, post: [post] + fts
    np.testing.assert_array_equal(np.array(col.geom_id_pairs), d["geom_id_pairs"])      # mink's constructor, pair for pair
    for k, t in enumerate(fts):
        t.set_target(mink.SE3(d["frame_targets"][:, k]))
    post = mink.PostureTask(m, cost=post_cost); post.set_target(d["posture_target"])
    tasks = tasks_order(fts, post)
    dt, damping = float(d["dt"]), float(d["damping"])
    G, h = col.compute_qp_inequalities(cfg, dt)
    n_box = d["h"].shape[1] - len(col.geom_id_pairs)
    h_ref = d["h"][:, n_box:] if name == "shadow_tips" else d["h"][:, -len(col.geom_id_pairs):]
    fin = np.isfinite(h_ref)
    assert (np.isfinite(h) == fin).all()
    np.testing.assert_allclose(h[fin], h_ref[fin], rtol=0, atol=1e-9 * max(1.0, np.abs(h_ref[fin]).max()))
    G_ref = d["G"][:, n_box:] if name == "shadow_tips" else d["G"][:, -len(col.geom_id_pairs):]
    mesh_rows = np.array([m.geom_type[a] == 7 or m.geom_type[b] == 7 for a, b in col.geom_id_pairs])
    nG = len(G_ref)
    np.testing.assert_allclose(G[:nG][:, ~mesh_rows], G_ref[:, ~mesh_rows], rtol=0, atol=1e-9)      # analytic pairs
    if mesh_rows.any():
        np.testing.assert_allclose(G[:nG][:, mesh_rows], G_ref[:, mesh_rows], rtol=0, atol=2e-5)  # GJK: direction to ~1e-6
    prob = mink.build_ik(cfg, tasks, dt, damping, lims)
    np.testing.assert_allclose(prob.P, d["H"], rtol=0, atol=1e-11 * np.abs(d["H"]).max())
    v, st = mink.solve_ik(cfg, tasks, dt, "mi355x", damping, limits=lims, return_status=True)
    assert (st & ~1 == 0).all(), st
    err = np.abs(v - d["v"]).max(axis=1) / np.maximum(1.0, np.abs(d["v"]).max(axis=1))
    print(name, "pairs", len(col.geom_id_pairs), "contacts per instance", fin.sum(axis=1).tolist(), "max rel v err %.2e" % err.max())
    assert err.max() < (1e-8 if not mesh_rows.any() else 5e-6)
