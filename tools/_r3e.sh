set -e
mkdir -p gpurun_out/r3e
python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3e/gen.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
for cfg in "256 2 8 4 4" "256 3 6 3 2" "256 3 6 4 1" "192 3 6 3 2" "384 3 6 3 2" "256 4 6 3 1" "128 4 6 3 1" "256 2 6 3 2" "256 2 5 3 3" "320 3 5 3 2"; do
  set -- $cfg
  for rep in 1 2; do
    dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/sw --timing --quiet --batchWindows $1 --computeThreads $2 --prepareThreads $3 --reduceThreads $4 --packThreads $5 | tail -1 | sed "s/^/batch=$1 compute=$2 prepare=$3 reduce=$4 pack=$5 :: /"
  done
done > gpurun_out/r3e/sweep.txt 2>&1
