#!/bin/bash
# Evidence for bench.py's `ragged` leg: tools/profile_ragged.sh TAG   (run on the GPU box from the repo root)
# Writes gpurun_out/prof_TAG/{ragged.json (un-profiled, per-launch ms), ragged_under_rocprof.json, kernel_stats.csv, pmc_<kernel>.json for the
# leg's two dominant kernels}.  PMC counters in their own runs with --kernel-trace only, FETCH_SIZE and WRITE_SIZE in separate passes.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
export TMPDIR=/tmp
DD_LAUNCH_TIMING=1 python3 $R/bench.py --ragged-only --steps 5 > $OUT/ragged.json
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --ragged-only --steps 5 > $OUT/ragged_under_rocprof.json 2> $OUT/trace.log
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
echo "trace done"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc_$i --output-format csv -- python3 $R/bench.py --ragged-only --steps 2 > $OUT/pmc_$i.json 2> $OUT/pmc_$i.log
  echo "pmc pass $i done"
done
# (since the full-length haplotypes run apart, the leg's K = 2 launches are the FOLDED builds <2, 6, *, true, 0, 1>)
python3 $R/tools/pmc_summary.py $OUT "dd_hmm_kernel<2, 6, false, true, 0, 1>" > $OUT/pmc_k2_lds.json
python3 $R/tools/pmc_summary.py $OUT "dd_hmm_kernel<5, 6, true, false, 0, 2>" > $OUT/pmc_k5_half.json
python3 $R/tools/pmc_summary.py $OUT "dd_hmm_kernel<2, 6, true, true, 0, 1>" > $OUT/pmc_k2_scratch.json
rm -rf $OUT/trace $OUT/pmc_[0-9] $OUT/*.log $OUT/pmc_[0-9].json
head -c 600 $OUT/pmc_k5_half.json
