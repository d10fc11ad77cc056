set -e
mkdir -p gpurun_out/r3g
python tools/concurrent_launch.py 256 > gpurun_out/r3g/concurrent.txt 2>&1
python tools/concurrent_launch.py 64 >> gpurun_out/r3g/concurrent.txt 2>&1
python tools/n2_pipeline_bench.py --windows 20000 --dir /tmp/n2b > gpurun_out/r3g/gen.txt 2>&1
export LD_LIBRARY_PATH=$PWD/dindel_tgi_amd/csrc:$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))'):/opt/rocm/lib
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl2 -- $R/dindel_tgi_amd/host/dindel_gpu --bamFile /tmp/n2b/reads.bam --varFile /tmp/n2b/windows.txt --hapFile /tmp/n2b/haps.txt --outputFile /tmp/n2b/tl --timing --quiet > $R/gpurun_out/r3g/run.txt 2>&1
ls -R /tmp/tl2 | head -20 > $R/gpurun_out/r3g/files.txt
python3 - <<'PY' > $R/gpurun_out/r3g/api_vs_kernel.txt 2>&1
import csv, glob
k = [r for f in glob.glob("/tmp/tl2/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
a = [r for f in glob.glob("/tmp/tl2/**/*hip_api_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
print("api columns:", list(a[0].keys()) if a else None)
hmm = sorted([r for r in k if "dd_hmm_kernel" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
launch = {r["Correlation_Id"]: r for r in a if "Launch" in r.get("Function", "")}
t0 = int(hmm[0]["Start_Timestamp"])
prev_end = None
for r in hmm[30:50]:
    l = launch.get(r["Correlation_Id"])
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("kernel q=%s start %.3f end %.3f ms; launch call at %s; previous kernel ended %.3f ms before this start; launch-to-start %.3f ms" % (
        r["Queue_Id"], (s - t0) / 1e6, (e - t0) / 1e6, "%.3f" % ((int(l["Start_Timestamp"]) - t0) / 1e6) if l else "?",
        (s - prev_end) / 1e6 if prev_end else 0.0, (s - int(l["Start_Timestamp"])) / 1e6 if l else -1))
    prev_end = e
PY
