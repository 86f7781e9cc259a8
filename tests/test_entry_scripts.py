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
    t = b.measured_traffic("g1_c3", 65536)
    assert t is None or t[0] > 0
    assert b.measured_traffic("g1_c3", 12345) is None
    assert b._free_port() > 0


def test_bench_refuses_wrong_world(monkeypatch):
    """--gpus N must never print a line from a different number of ranks (round-1 finding)."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout
