set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
{
for i in 1 2; do
  timeout 120 python tools/side_bench.py g1_full
  MKH_DEBUG_WAVES_PER_CU=9 timeout 120 python tools/side_bench.py g1_full
  MKH_DEBUG_NO_COM_W3=1 timeout 120 python tools/side_bench.py g1_full
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/c4_ab.log
cat gpurun_out/c4_ab.log
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc4; mkdir -p $O
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES --output-format csv -d $O/p1 -o k -- python $R/tools/pmc_workload.py 4 0 g1_full > $O/p1.log 2>&1
python $R/tools/rocprof_summary.py pmc $R/gpurun_out/pmc4_g1_full.json $O/p1 > /dev/null 2>&1
python - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc4_g1_full.json"))
ik=d["ik_solve_kernel"]; g=lambda c: sum(ik[c]["per_dispatch"])/len(ik[c]["per_dispatch"])
print(d["solve_kernel"], "waves", g("SQ_WAVES"), "wave_cyc/busy_cyc %.2f" % (g("SQ_WAVE_CYCLES")/g("SQ_BUSY_CYCLES")), "busy_cu/cycles %.2f" % (g("SQ_BUSY_CU_CYCLES")/g("SQ_CYCLES")), "valu share %.3f" % (g("SQ_ACTIVE_INST_VALU")/g("SQ_WAVE_CYCLES")))
PY
cd $R
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c4_tests.log 2>&1
cat gpurun_out/c4_tests.log
