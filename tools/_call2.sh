set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
echo "== g1_full: 3-waves F_COM build vs 2-waves"
for i in 1 2; do
  timeout 120 python tools/side_bench.py g1_full
  MKH_DEBUG_NO_COM_W3=1 timeout 120 python tools/side_bench.py g1_full
done
echo "== balanced grid (default) vs full grid"
for i in 1 2; do
  timeout 200 python tools/side_bench.py g1_c3:4096 g1_c3:8192 g1_c3:16384 g1_c3:24576 shadow_c4 g1_coll g1_plugin:16384 g1_full:16384
  MKH_DEBUG_NO_BALANCED_GRID=1 timeout 200 python tools/side_bench.py g1_c3:4096 g1_c3:8192 g1_c3:16384 g1_c3:24576 shadow_c4 g1_coll g1_plugin:16384 g1_full:16384
done
echo "== phase 0: next column's entry by readlane (rl) vs LDS broadcast"
for i in 1 2; do
  timeout 200 python tools/side_bench.py g1_plugin shadow_c4 g1_coll aloha_coll ur5e_c2:4096
  MKH_LIB_TAG=rl timeout 200 python tools/side_bench.py g1_plugin shadow_c4 g1_coll aloha_coll ur5e_c2:4096
done
} > gpurun_out/c2_ab.log 2>&1
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/c2_tests.log 2>&1
grep -v amdgpu.ids gpurun_out/c2_ab.log; cat gpurun_out/c2_tests.log
