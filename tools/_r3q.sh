set -e
mkdir -p gpurun_out/r3q
python -m pytest tests -x -q -m gpu -k "host or n2 or edge" > gpurun_out/r3q/pytest.log 2>&1 || { tail -30 gpurun_out/r3q/pytest.log; exit 1; }
python tools/n2_pipeline_bench.py --windows 60000 --dir /tmp/n2b > gpurun_out/r3q/gen.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
for z in 0 1 0 1 0 1; do
  DD_ZERO_COPY_IN=$z dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/z$z --timing --quiet | tail -1 | sed "s/^/zero_copy_in=$z :: /"
done > gpurun_out/r3q/ab.txt 2>&1
cmp /tmp/n2b/z0.glf.txt /tmp/n2b/z1.glf.txt && echo "identical output" >> gpurun_out/r3q/ab.txt
for z in 0 1; do
  DD_ZERO_COPY_IN=$z dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/f$z --timing --quiet --faster | tail -1 | sed "s/^/faster zero_copy_in=$z :: /"
done >> gpurun_out/r3q/ab.txt 2>&1
cmp /tmp/n2b/f0.glf.txt /tmp/n2b/f1.glf.txt && echo "identical output (faster)" >> gpurun_out/r3q/ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- $R/dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/tl --timing --quiet > $R/gpurun_out/r3q/run.txt 2>&1
python3 $R/tools/pipeline_timeline.py /tmp/tl > $R/gpurun_out/r3q/timeline.txt 2>&1
