#!/bin/bash
# Re-take the profiles/ evidence of a list of workloads on the binary in the tree, on ONE box:
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r05 g1_c3 g1_full shadow_c4'
# Per workload: the seven --pmc passes and the kernel-trace pass of tools/profile.sh (each its own rocprofv3 run), the summaries
# copied into profiles/ on the box so that the bench line of the same binary replays them, then that bench line
# (gpurun_out/<tag>_<config>_b<B>_bench.json; the CPU legs only for the headline).  Last: the default bench line with every
# other workload as `other_configs` (gpurun_out/<tag>_default_bench.json).  Copy gpurun_out/<tag>_* into profiles/ afterwards.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for CFG in "$@"; do
  B=$(python -c "import sys; sys.path.insert(0, '$R'); from mink_amd import workloads as w; print(w.BENCH_CONFIGS['$CFG']['batch'])")
  NAME=${CFG}_b${B}
  MKH_PROFILE_PMC_ONLY=1 bash tools/profile.sh "$TAG" "$CFG" > /dev/null 2>&1
  MKH_PROFILE_TRACE_ONLY=1 bash tools/profile.sh "$TAG" "$CFG" > /dev/null 2>&1
  cp "gpurun_out/${TAG}_${NAME}_pmc.json" "gpurun_out/${TAG}_${NAME}_kernel_stats.csv" profiles/ 2>/dev/null
  EXTRA="--no-cpu-baseline"; [ "$CFG" = g1_c3 ] && EXTRA=""
  timeout 300 python bench.py --config "$CFG" --batch "$B" --steps 20 --warmup 3 $EXTRA > "gpurun_out/${TAG}_${NAME}_bench.json" 2> "gpurun_out/${TAG}_${NAME}_bench.err"
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_${NAME}_bench.json")); r = d["roofline"]
    print("$CFG", "%.2f M/s" % (d["value"] / 1e6), r["kernel"], "kernel %.4f ms" % r["kernel_ms"], "traffic", r.get("traffic"), "stale", r.get("stale_profile"))
except Exception as e:
    print("$CFG", "FAILED", e)
PY
done
timeout 600 python bench.py > "gpurun_out/${TAG}_default_bench.json" 2> "gpurun_out/${TAG}_default_bench.err"
head -c 300 "gpurun_out/${TAG}_default_bench.json"; echo
