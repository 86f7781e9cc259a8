"""bench.py and __graft_entry__.py only run end-to-end on a GPU box; a NameError in a rarely executed
branch would surface there, at round end.  Static check on CPU: every global name their functions load is
defined at module level (or is a builtin), and the pure helpers behave."""

import ast
import builtins
import importlib.util
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _undefined_globals(path):
    tree = ast.parse(open(path).read())
    module_names = set(dir(builtins))
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
            module_names.add(node.name)
        elif isinstance(node, (ast.Import, ast.ImportFrom)):
            module_names.update((a.asname or a.name).split(".")[0] for a in node.names)
        elif isinstance(node, (ast.Assign, ast.AnnAssign, ast.AugAssign)):
            for t in (node.targets if isinstance(node, ast.Assign) else [node.target]):
                module_names.update(n.id for n in ast.walk(t) if isinstance(n, ast.Name))
        elif isinstance(node, (ast.If, ast.Try, ast.With, ast.For)):
            module_names.update(n.id for n in ast.walk(node) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store))
    missing = []
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Lambda))]:
        local = set()
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                local.add(n.id)
            elif isinstance(n, ast.arg):
                local.add(n.arg)
            elif isinstance(n, (ast.Import, ast.ImportFrom)):
                local.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n is not fn:
                local.add(n.name)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                local.add(n.name)
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in local and n.id not in module_names:
                missing.append((fn.name if hasattr(fn, "name") else "<lambda>", n.id, n.lineno))
    # nested functions see their enclosing function's locals: drop names defined anywhere in the file
    everywhere = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)} | \
                 {n.arg for n in ast.walk(tree) if isinstance(n, ast.arg)}
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            everywhere.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            everywhere.add(n.name)
    return [m for m in missing if m[1] not in everywhere]


@pytest.mark.parametrize("script", ["bench.py", "__graft_entry__.py", "tools/bench_configs.py", "tools/phase_profile.py",
                                    "tools/pmc_workload.py", "tools/rocprof_summary.py"])
def test_no_undefined_globals(script):
    assert _undefined_globals(os.path.join(REPO, script)) == []


def test_bench_helpers():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    from mink_amd import workloads
    assert workloads.BENCH_CONFIGS["g1_c3"]["bytes_per_solve"] == 924 and b.HBM_PEAK_GBS == 8000.0
    assert workloads.BENCH_CONFIGS["ur5e_c2"]["bytes_per_solve"] == 156
    assert workloads.BENCH_CONFIGS["shadow_c4"]["bytes_per_solve"] == 668
    assert b.issued_flop_per_solve("ik_solve_kernel_62_32_r44") > b.issued_flop_per_solve("ik_solve_kernel_8_0") > 0
    assert 1 <= b.usable_cpus() <= (os.cpu_count() or 1)
    assert b._free_port() > 0
    assert b.kernel_keys("ik_solve_kernel_48_72+redo_64") == ["ik_solve_kernel_48_72", "ik_solve_kernel_64_72"]
    assert b.kernel_keys("ik_quad_kernel") == ["ik_quad_kernel<8,0,16>"] and b.kernel_keys("ik_quad_kernel_32") == ["ik_quad_kernel<32,0,32>"] and b.kernel_keys("ik_lane_kernel_6") == ["ik_lane_kernel<6,0>"]
    assert b.kernel_keys("ik_solve_kernel_64_8+wide") == ["ik_solve_kernel_64_8", "ik_wide_kernel"]
    assert b.kernel_keys("ik_solve_kernel_48_8+redo_64+wide") == ["ik_solve_kernel_48_8", "ik_solve_kernel_64_8", "ik_wide_kernel"]
    assert b.kernel_keys("ik_solve_kernel_48_40_r48+redo_64+wide") == ["ik_solve_kernel_48_40_r48", "ik_solve_kernel_64_8", "ik_wide_kernel"]
    assert b.kernel_keys("convex_pre+ik_solve_kernel_16_8+wide") == ["convex_contacts_kernel", "ik_solve_kernel_16_8", "ik_wide_kernel_cvx"]
    assert b.kernel_keys("ik_solve_kernel_16_136+wide") == ["ik_solve_kernel_16_136", "ik_wide_kernel_cvx"]
    assert b.kernel_keys("ik_solve_kernel_48_72+redo_64+wide") == ["ik_solve_kernel_48_72", "ik_solve_kernel_64_72", "ik_wide_kernel"]


def test_counters_are_only_replayed_from_a_profile_of_the_code_that_ran(monkeypatch, tmp_path):
    """`roofline.traffic` / `valu_issue_view` come from profiles/ (counters cannot be read inside a timed run): a summary is
    used only when its provenance stamp (tools/rocprof_summary.py) names the kernels of this run with the sha256 of the
    code objects in the library now loaded (mink_amd/kernel_resources.json); anything else → None + stale_profile."""
    import json
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(REPO, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    table = b._resource_table()
    k = "ik_solve_kernel_44_32_r44_w3"
    assert table[k]["code_sha256"]
    monkeypatch.setattr(b, "REPO", str(tmp_path))
    monkeypatch.setattr(b, "_resource_table", lambda: table)
    (tmp_path / "profiles").mkdir()
    summ = {"hbm": {"traffic_bytes_per_launch": 64.7e6}, "derived": {"valu_active_per_wave_cycle": 0.25},
            "provenance": {"kernels": [k], "kernel_code_sha256": {k: table[k]["code_sha256"]}, "library_sha256": "x", "git_head": "y"}}
    path = tmp_path / "profiles" / "r99_g1_c3_b65536_pmc.json"
    path.write_text(json.dumps(summ))
    t, chk = b.measured_traffic("g1_c3", 65536, k)
    assert t == 64.7e6 and chk["kernel_code_match"] and not chk["stale_profile"]
    assert b.measured_valu_issue("g1_c3", 65536, 3.0, k)["simd_issue_slots_used"] == 0.75
    # another kernel ran than the one profiled / the kernel was rebuilt since / a summary without a stamp (rounds 1-3)
    t, chk = b.measured_traffic("g1_c3", 65536, "ik_solve_kernel_44_32_r44")
    assert t is None and chk["stale_profile"]
    summ["provenance"]["kernel_code_sha256"][k] = "0" * 64
    path.write_text(json.dumps(summ))
    t, chk = b.measured_traffic("g1_c3", 65536, k)
    assert t is None and chk["stale_profile"] and b.measured_valu_issue("g1_c3", 65536, 3.0, k) is None
    del summ["provenance"]
    path.write_text(json.dumps(summ))
    assert b.measured_traffic("g1_c3", 65536, k)[0] is None
    assert b.measured_traffic("g1_c3", 12345, k)[0] is None          # no summary of that batch at all


def test_bench_refuses_wrong_world(monkeypatch):
    """--gpus N must never print a line from a different number of ranks (round-1 finding)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout
