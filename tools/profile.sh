#!/bin/bash
# rocprofv3 evidence for profiles/: run on the GPU box through gpurun, e.g.
#   gpurun --timeout 900 -- 'bash tools/profile.sh r02a g1_c3'        (config defaults to g1_c3)
#   MKH_PROFILE_BATCH=1048576 bash tools/profile.sh r02b ur5e_c2      (a batch other than the config's own)
#   MKH_PROFILE_PMC_ONLY=1 bash tools/profile.sh r02a ur5e_c2         (the four counter passes only)
#   MKH_PROFILE_TRACE_ONLY=1 bash tools/profile.sh r05 shadow_c4      (the kernel-trace pass only)
# then copy gpurun_out/<tag>_* into profiles/.  Kernel trace/stats and each --pmc group are separate
# passes (counters are never combined with sys/runtime/hip traces).  Every rocprofv3 pass runs under a hard timeout:
# a profiler that fails to finalise (seen once after a GPU fault report inside the tool) must not eat the GPU budget.
set -u
TAG=${1:-rXX}
CFG=${2:-g1_c3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
B=${MKH_PROFILE_BATCH:-$(python -c "import sys; sys.path.insert(0, '$R'); from mink_amd import workloads as w; print(w.BENCH_CONFIGS['$CFG']['batch'])")}
NAME=${CFG}_b${B}
O=$R/gpurun_out/prof_${TAG}_$CFG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp

if [ -z "${MKH_PROFILE_PMC_ONLY:-}" ]; then
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o k -- \
  python "$R/bench.py" --config $CFG --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-pcie-leg --no-loop-legs > "$O/bench_under_trace.json" 2> "$O/trace.log"
python "$R/tools/rocprof_summary.py" stats "$R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv" "$O/trace" > /dev/null
fi

if [ -n "${MKH_PROFILE_TRACE_ONLY:-}" ]; then cp "$R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv" "$R/profiles/" 2>/dev/null; exit 0; fi
i=0
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" \
        "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
        "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
        "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
        "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
        "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES")
DIRS=()
for grp in "${PMC_GROUPS[@]}"; do
  i=$((i + 1))
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d "$O/pmc$i" -o k -- \
    python "$R/tools/pmc_workload.py" 4 $B $CFG > "$O/pmc$i.log" 2>&1
  DIRS+=("$O/pmc$i")
done
python "$R/tools/rocprof_summary.py" pmc "$R/gpurun_out/${TAG}_${NAME}_pmc.json" "${DIRS[@]}"
if [ -z "${MKH_PROFILE_PMC_ONLY:-}" ]; then
# the plain bench line of the same binary on the same box replays the counters just collected (bench.py reads profiles/ and
# checks the summary's code-object hashes against what it runs)
cp "$R/gpurun_out/${TAG}_${NAME}_pmc.json" "$R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv" "$R/profiles/" 2>/dev/null
python "$R/bench.py" --config $CFG --batch $B --steps 20 --warmup 3 > "$R/gpurun_out/${TAG}_${NAME}_bench.json" 2> "$O/bench.err"
cat "$R/gpurun_out/${TAG}_${NAME}_bench.json"
head -5 "$R/gpurun_out/${TAG}_${NAME}_kernel_stats.csv"
fi
