set -e
mkdir -p gpurun_out/r3s
python tools/n2_pipeline_bench.py --windows 60000 --dir /tmp/n2b > gpurun_out/r3s/gen.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
for cfg in "2 4 8 4" "3 2 6 4" "3 2 8 4" "3 3 6 3" "4 1 6 4" "3 2 6 4" "2 4 8 4"; do set -- $cfg
  dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/e --timing --quiet --computeThreads $1 --packThreads $2 --prepareThreads $3 --reduceThreads $4 | tail -1 | sed "s/^/engines=$1 pack=$2 prepare=$3 reduce=$4 :: /"
done > gpurun_out/r3s/engines.txt 2>&1
