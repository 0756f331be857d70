#!/bin/bash
# rocprofv3 profiles of the default bench command (run on the GPU box through gpurun):
#   tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/...   then   tools/summarize_profile.py <tag>
# --kernel-trace --stats and every --pmc group run as SEPARATE passes (never combined).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $REPO/bench.py --steps 10 --warmup 2 --cpu-sample 4096 --sustained 0"   # bench.py's own default step counts
# the default command (verify + x25519 + sign in one process), then each op alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_all -- $CMD > $OUT/bench_stats_all.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD --op verify > $OUT/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_x25519 -- $CMD --op x25519 > $OUT/bench_stats_x25519.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_sign -- $CMD --op sign > $OUT/bench_stats_sign.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD --op verify > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD --op verify > $OUT/bench_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_x25519 -- $CMD --op x25519 > $OUT/bench_fetch_x.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_x25519 -- $CMD --op x25519 > $OUT/bench_write_x.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_sign -- $CMD --op sign > $OUT/bench_fetch_s.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_sign -- $CMD --op sign > $OUT/bench_write_s.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- $CMD --op verify > $OUT/bench_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $OUT/pmc_misc -- $CMD --op verify > $OUT/bench_misc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq_x25519 -- $CMD --op x25519 > $OUT/bench_sq_x.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq_sign -- $CMD --op sign > $OUT/bench_sq_s.log 2>&1
# the worst case (tools/exact_lane_probe.py: every item through the exact path, then genuine signatures under random keys): the one-lane
# exact kernels' own durations, and their counters from the passes in which they have the chip to themselves ("alone": self-check mode 2 -
# beside k_verify_main_half a kernel's cycles are not its own)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_exact -- python3 $REPO/tools/exact_lane_probe.py 20 3 > $OUT/bench_stats_exact.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq_exact -- python3 $REPO/tools/exact_lane_probe.py 20 3 alone > $OUT/bench_sq_exact.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_exact -- python3 $REPO/tools/exact_lane_probe.py 20 3 alone > $OUT/bench_fetch_exact.log 2>&1
# the opt-in batch verification (tools/rlc_rate.py): kernel stats, then its counters (the k_rlc_* kernels' rows of the summary)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_rlc -- python3 $REPO/tools/rlc_rate.py 5 > $OUT/bench_stats_rlc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq_rlc -- python3 $REPO/tools/rlc_rate.py 3 > $OUT/bench_sq_rlc.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_rlc -- python3 $REPO/tools/rlc_rate.py 3 > $OUT/bench_fetch_rlc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_rlc -- python3 $REPO/tools/rlc_rate.py 3 > $OUT/bench_write_rlc.log 2>&1
python3 $REPO/tools/summarize_profile.py $TAG
