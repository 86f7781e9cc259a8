#!/bin/bash
# Dynamic instruction census of a solve kernel BY PHASE x CLASS: PMC passes over launches of the clock build (library tag `clk`,
# -DMKH_CLOCKS) that abandon every solve after phase boundary k (MKH_DEBUG_PHASE_STOP=k, ik_kernel.h MKH_STOP); the counters of
# stop = k minus those of stop = k - 1 are phase k's instructions.  Boundaries of the low-rank W3 builds:
#   1 FK  2 joint axes / dof lanes  3 task lanes  4 posture / damping  5 box limits  6 Jacobian rows  7 S and w
#   8 eliminations  9 rank-1 updates of the dof block (tableau built)  0 = whole solve (active set + write-back after 9)
#     MKH_BUILD_TAG=clk MKH_EXTRA_FLAGS=-DMKH_CLOCKS python -m mink_amd.csrc.build       # build container
#     gpurun -- 'bash tools/phase_census.sh r04 g1_c3'                                    # -> gpurun_out/r04_g1_c3_phase_census.{json,md}
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}; CFG=${2:-g1_c3}
O=$R/gpurun_out/census_${TAG}_$CFG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
         "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD")
for stop in 1 2 3 4 5 6 7 8 9 0; do
  DIRS=()
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i + 1))
    MKH_LIB_TAG=clk MKH_DEBUG_CLOCKS=/tmp/mkh_census_clk.bin MKH_DEBUG_PHASE_STOP=$stop timeout -s KILL 180 \
      rocprofv3 --pmc $grp --output-format csv -d $O/s${stop}_$i -o k -- python $R/tools/pmc_workload.py 2 0 $CFG > $O/s${stop}_$i.log 2>&1
    DIRS+=("$O/s${stop}_$i")
  done
  python $R/tools/rocprof_summary.py pmc $O/stop$stop.json "${DIRS[@]}" > /dev/null 2>&1
done
python $R/tools/phase_census.py $O $R/gpurun_out/${TAG}_${CFG}_phase_census $CFG
