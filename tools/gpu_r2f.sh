#!/bin/bash
# PMC passes over the backward (tools/pmc_probe.py --bwd) with the XCD-affine apply on and off
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for X in 1 0; do
  TAG=r2f_xcd$X
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"; do
    i=$((i+1))
    PARAM_AMD_BWD_XCD=$X PARAM_AMD_BWD_PHASES=1 timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${TAG}_pmc/pmc_$i -o pmc -- \
        python $GRAFT_REPO_ROOT/tools/pmc_probe.py --bwd --manifest $OUT/${TAG}_pmc/pmc_manifest.json > $OUT/${TAG}_pmc_$i.log 2>&1
  done
  ls $OUT/${TAG}_pmc
done
