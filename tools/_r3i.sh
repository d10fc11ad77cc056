set -e
mkdir -p gpurun_out/r3i
python -m pytest tests -x -q -m gpu > gpurun_out/r3i/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r3i/pytest_gpu.log; exit 1; }
for env in "DD_NO_FOLD=1" "DD_X=1"; do
  for shape in "100 8 200 120 5 6000" "100 8 200 117 5 6000" "100 8 200 121 5 6000" "100 8 200 123 5 6000" "100 8 200 56 5 6000" "100 8 200 120 10 4000" "250 16 200 120 10 500"; do
    env $env python tools/ab_point.py $shape 2>/dev/null | sed "s/^/$env /"
  done
done > gpurun_out/r3i/fold_ab.txt 2>&1
python bench.py --steps 5 --warmup 1 --kernel-only --no-cpu-baseline > gpurun_out/r3i/bench.json 2>/dev/null
