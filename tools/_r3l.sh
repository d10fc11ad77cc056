set -e
mkdir -p gpurun_out/r3l
python -m pytest tests -x -q -m gpu > gpurun_out/r3l/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r3l/pytest_gpu.log; exit 1; }
timeout -k 10 420 python tools/fuzz_campaign.py --seconds 360 --seed0 900000 > gpurun_out/r3l/fuzz.txt 2>&1
