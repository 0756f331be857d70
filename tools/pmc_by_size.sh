#!/bin/bash
# Counters of one verify pass by size (VERDICT r03 #1a): L2 hits / misses, fabric read / write requests, clock, issue stalls,
# per kernel, each group its own rocprofv3 run:   tools/pmc_by_size.sh <tag> [sizes...]   -> gpurun_out/pmc_<tag>/summary.txt
TAG=${1:-sz}; shift
SIZES=${@:-17 18 20}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|MALL[A-Z0-9_]*\|[A-Z_]*EA[0-9]*_[A-Z0-9_]*" | sort -u > $OUT/counter_names.txt
for L in $SIZES; do
  for KIND in mix valid; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_${L}_$KIND -- python3 $REPO/tools/verify_pass.py $L 10 $KIND > $OUT/run_stats_${L}_$KIND.log 2>&1
  done
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc_$L -- python3 $REPO/tools/verify_pass.py $L 5 mix > $OUT/run_tcc_$L.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_$L -- python3 $REPO/tools/verify_pass.py $L 5 mix > $OUT/run_sq_$L.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/tcc2_$L -- python3 $REPO/tools/verify_pass.py $L 5 mix > $OUT/run_tcc2_$L.log 2>&1
done
python3 $REPO/tools/pmc_by_size_summary.py $OUT $SIZES | tee $OUT/summary.txt
