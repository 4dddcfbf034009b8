#!/bin/bash
# Regenerates EVERY artefact under profiles/ for one round from the tree it runs in
# (VERDICT r2 item 2).  Run on the GPU box through gpurun:
#
#   gpurun --timeout 3000 -- 'bash tools/regen_profiles.sh r03 <commit>'
#
# For each workload: one rocprofv3 --kernel-trace --stats run, then SEPARATE --pmc
# passes (SQ issue / stall counters, instruction mix, LDS, FETCH_SIZE, WRITE_SIZE --
# the guide's HBM recipe: FETCH_SIZE and WRITE_SIZE cannot share a pass), the program
# directly after `--`.  Raw output goes to gpurun_out/<round>_<workload>/, the
# condensed files (tools/summarize_pmc.py) to gpurun_out/profiles_<round>/, from where
# they are copied into profiles/ and committed.  Every summary carries the commit.
set -u
ROUND=${1:-r03}
COMMIT=${2:-unknown}
shift; shift
ONLY=${@:-c2 c3 c4 dense64 structural structural_ar general adaptive logit pg probit xtx_c2 xtx_c4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
DEST=$ROOT/gpurun_out/profiles_$ROUND
mkdir -p $DEST
cd /tmp && export TMPDIR=/tmp

profile() {   # name, pmc (yes|no), kernel substrings for the summary, then the python script + args
  local name=$1 pmc=$2 keys=$3; shift 3
  local out=$ROOT/gpurun_out/${ROUND}_$name
  mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 "$@" > $out/stats.out 2> $out/stats.err
  if [ "$pmc" = yes ]; then
    pass() {
      local pn=$1; shift
      rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_$pn -o pmc -- python3 "${CMD[@]}" > $out/pmc_$pn.out 2> $out/pmc_$pn.err
    }
    CMD=("$@" ${PMC_EXTRA:-})   # (PMC_EXTRA: flags for the counter passes only)
    pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
    pass sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
    pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
    pass fetch FETCH_SIZE
    pass write WRITE_SIZE
    # (round 6, VERDICT r5 weak 6: what binds the f64 GEMMs, shown by a counter)
    pass mfma SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  fi
  python3 $ROOT/tools/summarize_pmc.py $out $DEST/${ROUND}_$name --commit $COMMIT --command "$*" $keys > /dev/null
  tail -2 $out/stats.out
}

for w in $ONLY; do
  case $w in
    c2)         PMC_EXTRA="--keep-apart" profile c2 yes "ssvs_ xtx_mfma plane_sum col_reduce" $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-curve ;;
    c3)         profile c3 yes "ss_round ssvs_ kalman xtwx_" $ROOT/tools/ss_bench.py ;;
    structural) profile structural yes "ssvs_ ssm_ xtwx_" $ROOT/tools/structural_bench.py 2,12,1024 ;;
    structural_ar) profile structural_ar no "ssvs_ ssm_ xtwx_" $ROOT/tools/structural_bench.py 2,12,1024,2 ;;
    c4)         profile c4 yes "ssvs_ xtx_mfma plane_sum col_reduce" $ROOT/bench.py --config 3 --steps 3 --warmup 1 ;;
    dense64)    export NSWEEP=1000; profile dense64 yes "ssvs_" $ROOT/tools/dense_variant.py 64; unset NSWEEP ;;
    general)    profile general yes "ssvs_ ssg_ ssm_ xtwx_" $ROOT/tools/ssg_bench.py ;;
    adaptive)   profile adaptive yes "ssvs_" $ROOT/tools/adaptive_bench.py ;;
    pg)         profile pg yes "ssvs_ logit_ xtwx_ plain_reduce" $ROOT/tools/probit_bench.py 50000 1024 8 512 pg 12 ;;
    logit)      profile logit yes "ssvs_ logit_ xtwx_ plain_reduce" $ROOT/tools/probit_bench.py 50000 1024 8 512 logit 12 ;;
    probit)     profile probit yes "ssvs_ probit_ xtwx_ plain_reduce" $ROOT/tools/probit_bench.py 50000 1024 8 512 probit 12 ;;
    xtx_c2)     profile xtx_c2 no "xtx_mfma plane_sum col_reduce" $ROOT/tools/suf_bench.py 10000 512 20 ;;
    xtx_c4)     profile xtx_c4 yes "xtx_mfma plane_sum col_reduce" $ROOT/tools/suf_bench.py 100000 4096 5 ;;
  esac
done
ls -la $DEST
