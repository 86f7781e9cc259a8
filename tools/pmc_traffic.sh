#!/bin/bash
# usage: pmc2.sh <libtag-or-empty> <config>
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; CFG=$2
O=$R/gpurun_out/pmc2_${TAG:-default}_$CFG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  MKH_LIB_TAG=$TAG timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d $O/pmc$i -o k -- python $R/tools/pmc_workload.py 4 0 $CFG > $O/pmc$i.log 2>&1
done
python $R/tools/rocprof_summary.py pmc $R/gpurun_out/pmc2_${TAG:-default}_${CFG}.json $O/pmc1 $O/pmc2 | tail -12
