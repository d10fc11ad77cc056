#!/bin/bash
# VALU counters of one (shape, build) point: tools/pmc_point.sh TAG L H R hap mld windows   (env as for tools/ab_point.py; run on the GPU box)
# -> gpurun_out/pmc_TAG.json: counters averaged per dispatch of dd_hmm_kernel, instructions per pair
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/pmcpt_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc --output-format csv -- python3 $R/tools/ab_point.py "$@" > $OUT/point.txt 2> $OUT/pmc.log
python3 $R/tools/pmc_summary.py $OUT dd_hmm_kernel > $R/gpurun_out/pmc_$TAG.json
grep -v amdgpu $OUT/point.txt >> $R/gpurun_out/pmc_$TAG.json
rm -rf $OUT
