set -e
mkdir -p gpurun_out/r3z
python -m pytest tests -x -q -m gpu > gpurun_out/r3z/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r3z/pytest_gpu.log; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3z/smoke.txt 2>&1
bash tools/profile_round.sh d3 dd_hmm_kernel > gpurun_out/r3z/profile_d3.log 2>&1
bash tools/profile_round.sh f3 dd_faster_kernel --faster > gpurun_out/r3z/profile_f3.log 2>&1
