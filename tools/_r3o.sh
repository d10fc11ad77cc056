set -e
mkdir -p gpurun_out/r3o
python -m pytest tests -x -q -m gpu > gpurun_out/r3o/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/r3o/pytest_gpu.log; exit 1; }
for env in "DD_NO_FOLD=1" "DD_X=1"; do
  for shape in "100 8 200 120 5 6000" "100 8 200 121 5 6000" "100 8 200 122 5 6000" "100 8 200 123 5 6000"; do
    env $env python tools/ab_point.py $shape 2>/dev/null | sed "s/^/$env /"
  done
done > gpurun_out/r3o/fold_ab2.txt 2>&1
python tools/n2_pipeline_bench.py --windows 40000 --dir /tmp/n2b > gpurun_out/r3o/n2_40k.jsonl 2>&1
AB_DIRS=$PWD/tools/_ab/r2end_host bash tools/n2_ab_compare.sh /tmp/n2b > gpurun_out/r3o/ab.txt 2>&1
timeout -k 10 200 python tools/fuzz_campaign.py --seconds 150 --seed0 950000 > gpurun_out/r3o/fuzz.txt 2>&1
