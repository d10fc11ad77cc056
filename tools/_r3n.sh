set -e
mkdir -p gpurun_out/r3n
python -m pytest tests -x -q -m gpu > gpurun_out/r3n/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r3n/pytest_gpu.log; exit 1; }
python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3n/n2_40k.jsonl 2>&1
AB_DIRS=$PWD/tools/_ab/r2end_host bash tools/n2_ab_compare.sh /tmp/n2b > gpurun_out/r3n/ab.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- $R/dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/tl --timing --quiet > $R/gpurun_out/r3n/run.txt 2>&1
python3 $R/tools/pipeline_timeline.py /tmp/tl 1000 1030 > $R/gpurun_out/r3n/timeline.txt 2>&1
