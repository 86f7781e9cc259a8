set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
GROUPS_=("FETCH_SIZE" "WRITE_SIZE" \
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
 "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
 "SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_IFETCH SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES")
for mode in w3 w2; do
  O=$R/gpurun_out/pmc3_$mode
  mkdir -p $O
  DIRS=()
  i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1))
    if [ $mode = w2 ]; then export MKH_DEBUG_NO_COM_W3=1; else unset MKH_DEBUG_NO_COM_W3; fi
    timeout -s KILL 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc$i -o k -- python $R/tools/pmc_workload.py 4 0 g1_full > $O/pmc$i.log 2>&1
    DIRS+=("$O/pmc$i")
  done
  python $R/tools/rocprof_summary.py pmc $R/gpurun_out/pmc3_g1_full_$mode.json "${DIRS[@]}" > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc3_g1_full_$mode.json"))
print("$mode", d["solve_kernel"])
print(json.dumps(d["derived"], indent=0))
print(json.dumps({k:v for k,v in d["hbm"].items() if not isinstance(v,dict)}))
print({k:(sum(v["per_dispatch"])/len(v["per_dispatch"])) for k,v in d["ik_solve_kernel"].items()})
print(d["kernel_resources"])
PY
done
