set -e
mkdir -p gpurun_out/r3d
python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3d/gen.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- $R/dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/tl --timing --quiet > $R/gpurun_out/r3d/run.txt 2>&1
python3 $R/tools/pipeline_timeline.py /tmp/tl 1000 1060 > $R/gpurun_out/r3d/timeline.txt 2>&1
