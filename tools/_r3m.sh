set -e
mkdir -p gpurun_out/r3m
hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o /tmp/valu_rate && timeout -k 10 180 /tmp/valu_rate > gpurun_out/r3m/valu_rate.txt 2>&1
bash tools/profile_round.sh d3 dd_hmm_kernel > gpurun_out/r3m/profile_round.log 2>&1
python tools/stress_sweep.py > gpurun_out/r3m/stress_sweep.jsonl 2> gpurun_out/r3m/stress.err
python bench.py --steps 5 --warmup 1 --kernel-only --no-cpu-baseline --max-length-del 10 > gpurun_out/r3m/bench_mld10.json 2>/dev/null
