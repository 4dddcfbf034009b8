#!/bin/bash
# Profiles of the headline bench on the GPU box (run through gpurun):
#   kernel-trace + stats, then separate PMC passes (SQ issue/stall counters, LDS,
#   FETCH_SIZE, WRITE_SIZE).  Outputs under gpurun_out/$TAG/; the summaries that
#   are to be judged are copied into profiles/ by hand afterwards.
# usage: tools/profile_bench.sh TAG [script args...]   (default script: bench.py)
set -u
TAG=${1:-r02}
shift || true
SCRIPT=${PROFILE_SCRIPT:-bench.py}
ARGS=${@:---no-cpu-baseline --no-curve}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/$SCRIPT $ARGS > $OUT/stats.json 2> $OUT/stats.err
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $ROOT/$SCRIPT $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
find $OUT -name "*.csv" | head -40
