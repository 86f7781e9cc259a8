set -u
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
{
for i in 1 2; do
  timeout 120 python tools/side_bench.py g1_full aloha_coll
  MKH_DEBUG_STATIC_78=1 timeout 120 python tools/side_bench.py g1_full aloha_coll
  MKH_DEBUG_NO_COM_W3=1 MKH_DEBUG_WAVES_PER_CU=4 timeout 120 python tools/side_bench.py g1_full aloha_coll
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/c5_ab.log
cat gpurun_out/c5_ab.log
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c5_tests.log 2>&1
cat gpurun_out/c5_tests.log
