#!/bin/bash
# rocprofv3 evidence for profiles/: run on the GPU box through gpurun, e.g.
#   gpurun --timeout 900 -- 'bash tools/profile.sh r01c'
# then copy gpurun_out/<tag>_* into profiles/.  Kernel trace/stats and each --pmc group are separate
# passes (counters are never combined with sys/runtime/hip traces).  Every rocprofv3 pass runs under a hard timeout:
# a profiler that fails to finalise (seen once after a GPU fault report inside the tool) must not eat the GPU budget.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp

timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o g1 -- \
  python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$O/bench_under_trace.json" 2> "$O/trace.log"
python "$R/tools/rocprof_summary.py" stats "$R/gpurun_out/${TAG}_g1_b65536_kernel_stats.csv" "$O/trace" > /dev/null

i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  i=$((i + 1))
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d "$O/pmc$i" -o g1 -- \
    python "$R/tools/pmc_workload.py" 4 > "$O/pmc$i.log" 2>&1
done
python "$R/tools/rocprof_summary.py" pmc "$R/gpurun_out/${TAG}_g1_b65536_pmc.json" "$O"/pmc1 "$O"/pmc2 "$O"/pmc3 "$O"/pmc4
python "$R/bench.py" --steps 20 --warmup 3 > "$R/gpurun_out/${TAG}_bench.json"
cat "$R/gpurun_out/${TAG}_bench.json"
head -5 "$R/gpurun_out/${TAG}_g1_b65536_kernel_stats.csv"
