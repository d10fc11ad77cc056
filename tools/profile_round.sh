#!/bin/bash
# Round-end evidence for profiles/: tools/profile_round.sh TAG KERNEL_SUBSTRING [bench.py args...]
# Run on the GPU box from the repo root.  Writes gpurun_out/prof_TAG/{bench.json, bench_under_rocprof.json,
# kernel_stats.csv, pmc.json}.  PMC counters are collected in their own runs with --kernel-trace only, FETCH_SIZE and
# WRITE_SIZE in separate passes; the profiled runs time the kernel path only (--kernel-only: the host-API and C++ adapter legs
# launch the same kernel on sub-batches and would blur the per-launch averages) (they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; KSUB=$2; shift 2
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
python3 $R/bench.py --steps 5 --warmup 1 $BENCH_EXTRA "$@" > $OUT/bench.json      # BENCH_EXTRA=--host-api: un-profiled line only
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --kernel-only --no-ragged "$@" > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
echo "trace done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$i --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --kernel-only --no-ragged "$@" > $OUT/pmc_$i.json 2> $OUT/pmc_$i.log
  echo "pmc pass $i done"
done
python3 $R/tools/pmc_summary.py $OUT "$KSUB" $OUT/bench.json > $OUT/pmc.json
python3 $R/tools/refresh_traffic.py $OUT/bench.json $OUT/pmc.json "profiles/<round>/<tag>_pmc.json" || true   # the line was written before these passes
rm -rf $OUT/trace $OUT/pmc_[0-9] $OUT/*.log
cat $OUT/pmc.json
