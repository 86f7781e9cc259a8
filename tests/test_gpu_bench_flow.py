"""bench.py's own multi-rank control flow on the 1-GPU box: `python bench.py --gpus 2` launches its two ranks itself
(torch.distributed.run, 127.0.0.1), MKH_BENCH_SHARE_GPU=1 lets both use device 0 over gloo (RCCL refuses two ranks on
one device) — timings are meaningless, the flow is what is tested: self-launch, barriers, MAX-reduced time, the compute-only
region and the gather region in ONE line whose n_gpus equals --gpus; and without the hook a 2-GPU line is refused."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_self_launched_two_rank_line():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4096", "--no-cpu-baseline"], env=_env(MKH_BENCH_SHARE_GPU="1"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["global_batch"] == 2 * 4096
    assert d["value"] > 0 and d["gather"]["value"] > 0 and d["gather"]["bytes_per_step_to_rank0"] == 4096 * 43 * 8
    assert "roofline" in d and d["scaling"] == "weak" and "cpu_baseline" not in d
    # what the process group saw: its own world size and every rank's device (here: both ranks on device 0 — the line says so)
    c = d["config"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and [x["rank"] for x in c["devices"]] == [0, 1]
    assert all(x["name"] and x["pci_bus_id"] for x in c["devices"]) and c["distinct_devices"] == 1
    # the counters of the line are either a profile of the code that ran, or absent and flagged
    rf = d["roofline"]
    assert rf["stale_profile"] == (rf["traffic"] is None) and rf["profile_check"]["kernels_ran"]


def test_two_gpu_line_is_refused_on_one_gpu():
    from mink_amd import _native as nat
    if nat.lib().mkh_device_count() >= 2:
        pytest.skip("box has two GPUs")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "only 1 GPU(s) visible" in r.stderr and "{" not in r.stdout


def test_rccl_process_group_of_one_runs_the_gather_and_the_timing_reduce():
    """The collectives of the N-rank bench — `gather_rows` (dist.gather of v into row blocks of a preallocated result) and the
    MAX all-reduce of the elapsed time — on the `nccl` backend (= RCCL), in the only world this box can form: one rank.  It
    proves the RCCL build initialises on the box and that the calls the scaling run makes execute on device tensors; it says
    nothing about xGMI."""
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from mink_amd.distributed import gather_rows, shard_bounds
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", world_size=1, rank=0)
assert dist.get_backend() == "nccl"
v = torch.arange(4096 * 43, dtype=torch.float64, device="cuda").reshape(4096, 43)
out = torch.empty_like(v)
got = gather_rows(v, 4096, dst=0, out=out)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert got.data_ptr() == out.data_ptr() and torch.equal(out, v) and t.item() == 1.25 and shard_bounds(4096, 1, 0) == (0, 4096)
dist.destroy_process_group()
print("RCCL_OK")
""" % REPO
    r = subprocess.run([sys.executable, "-c", code], env=_env(HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
