set -e
mkdir -p gpurun_out/r3r
python tools/n2_pipeline_bench.py --windows 20000 --dir /tmp/n2b > gpurun_out/r3r/gen.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
DD_TIMING=1 dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/t --timing --quiet > gpurun_out/r3r/timing.txt 2>&1
