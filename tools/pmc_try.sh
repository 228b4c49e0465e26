#!/bin/bash
export TMPDIR=/tmp
cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "FETCH_SIZE|TCC_EA0_RDREQ|TCC_EA_RDREQ|TCC_BUBBLE|WRITE_SIZE|TCC_EA0_WRREQ" | head -20
cd $GRAFT_REPO_ROOT
for C in "TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum" ; do
  rm -rf gpurun_out/pmc_try_$C
  SVA_CONCURRENCY=0 timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_try_$C -o p -- python bench.py --no-cpu-baseline --no-roofline --no-graph --steps 4 --warmup 1 > gpurun_out/pmc_try_$C.log 2>&1
  echo "rc=$? for $C"; ls gpurun_out/pmc_try_$C 2>/dev/null
done
