set -e
mkdir -p gpurun_out/r3j
python tools/n2_pipeline_bench.py --windows 10000 --dir /tmp/n2 > gpurun_out/r3j/n2_10k.jsonl 2>&1
AB_DIRS=$PWD/tools/_ab/r2end_host bash tools/n2_ab_compare.sh /tmp/n2 > gpurun_out/r3j/ab_uniform.txt 2>&1
python tools/n2_pipeline_bench.py --windows 8000 --ragged --dir /tmp/n2r > gpurun_out/r3j/n2_ragged.jsonl 2>&1
AB_DIRS=$PWD/tools/_ab/r2end_host bash tools/n2_ab_compare.sh /tmp/n2r > gpurun_out/r3j/ab_ragged.txt 2>&1
python -m pytest tests -x -q -m gpu -k "n2 or host or genotype or glf" > gpurun_out/r3j/pytest.log 2>&1 || true
python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3j/n2_40k.jsonl 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
DINDEL_REDUCE_TIMING=1 dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/rt --timing --quiet > gpurun_out/r3j/reduce_timing.txt 2>&1
