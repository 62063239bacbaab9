#!/bin/bash
# PMC passes of the hot kernels (tools/prof_hot2.py), one rocprofv3 run per counter group, as the
# microarchitecture guide prescribes (no tracing domains besides --kernel-trace next to --pmc).
# usage (on the GPU box): tools/pmc_collect.sh <out-dir under gpurun_out/> ; then tools/pmc_parse.py <dir> > profiles/roundN_pmc.json
set -e
out=gpurun_out/$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pass_$i -- python3 tools/prof_hot2.py > $out/pass_$i.log 2>&1 || { tail -5 $out/pass_$i.log; exit 1; }
  echo "pass $i ($grp) done"
done
