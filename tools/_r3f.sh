set -e
mkdir -p gpurun_out/r3f
for n in 128 256 384 512 768 1024 2048; do
  for env in "DD_SPLIT_NO_ROUNDS=1" "DD_X=1"; do
    env $env python tools/ab_point.py 100 8 200 120 5 $n 2>/dev/null | sed "s/^/$env /"
  done
done > gpurun_out/r3f/split_ab.txt 2>&1
python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3f/n2_40k.jsonl 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
for cfg in "256 2 8 4 4" "256 3 6 4 1" "512 2 8 4 4"; do
  set -- $cfg
  for rep in 1 2; do
    dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/sw --timing --quiet --batchWindows $1 --computeThreads $2 --prepareThreads $3 --reduceThreads $4 --packThreads $5 | tail -1 | sed "s/^/batch=$1 compute=$2 prepare=$3 reduce=$4 pack=$5 :: /"
  done
done > gpurun_out/r3f/sweep.txt 2>&1
AB_DIRS=$PWD/tools/_ab/r2end_host bash tools/n2_ab_compare.sh /tmp/n2b > gpurun_out/r3f/ab.txt 2>&1
