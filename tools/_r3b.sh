set -e
mkdir -p gpurun_out/r3b
python tools/n2_pipeline_bench.py --windows 10000 --dir /tmp/n2 > gpurun_out/r3b/n2_10k.jsonl 2>&1
AB_DIRS=$PWD/tools/_ab/r2end_host bash tools/n2_ab_compare.sh /tmp/n2 > gpurun_out/r3b/ab_uniform.txt 2>&1
python tools/n2_pipeline_bench.py --windows 8000 --ragged --dir /tmp/n2r > gpurun_out/r3b/n2_ragged.jsonl 2>&1
AB_DIRS=$PWD/tools/_ab/r2end_host bash tools/n2_ab_compare.sh /tmp/n2r > gpurun_out/r3b/ab_ragged.txt 2>&1
DINDEL_REDUCE_TIMING=1 python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3b/n2_40k.jsonl 2> gpurun_out/r3b/n2_40k.err
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
for pt in 1 2 4 8 12; do dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/po --prepareOnly --timing --quiet --prepareThreads $pt | tail -1; done > gpurun_out/r3b/prepare_scaling.txt 2>&1
DINDEL_REDUCE_TIMING=1 dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/rt --timing --quiet > gpurun_out/r3b/reduce_timing.txt 2>&1
nproc >> gpurun_out/r3b/prepare_scaling.txt
