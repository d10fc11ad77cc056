set -e
mkdir -p gpurun_out/r3p
python tools/n2_pipeline_bench.py --windows 60000 --dir /tmp/n2b > gpurun_out/r3p/gen.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
for ahead in 3 8 16; do for rep in 1 2 3; do
  dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/sw --timing --quiet --computeAhead $ahead | tail -1 | sed "s/^/ahead=$ahead :: /"
done; done > gpurun_out/r3p/ahead.txt 2>&1
for cfg in "8 3" "16 3"; do set -- $cfg; for rep in 1 2; do
  dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/sw --timing --quiet --computeAhead $1 --computeThreads $2 --packThreads 2 | tail -1 | sed "s/^/ahead=$1 engines=$2 :: /"
done; done >> gpurun_out/r3p/ahead.txt 2>&1
