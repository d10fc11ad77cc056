set -e
mkdir -p gpurun_out/r3h
python -m pytest tests -x -q -m gpu > gpurun_out/r3h/pytest_gpu.log 2>&1
python tools/n2_pipeline_bench.py --windows 100000 --dir /tmp/n2c > gpurun_out/r3h/n2_100k.jsonl 2>&1
python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3h/n2_40k.jsonl 2>&1
