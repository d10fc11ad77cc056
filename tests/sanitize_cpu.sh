#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (oracle restatement, C++ host adapter): GPU sanitizers are not
# available on the pool, so the device code is covered by parity tests only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -fPIC -std=c99 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -o /tmp/libdd_oracle_asan.so "$ROOT/oracle/dd_oracle.c" -lm
g++ -O1 -g -fPIC -std=c++11 -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o /tmp/libdindel_host_asan.so \
    "$ROOT"/dindel_tgi_amd/host/{compute_likelihoods,genotype,cigar,host_capi,glf_to_vcf,bam_reader,window_io,get_reads,diploid_glf,realigned_bam,fast_inflate}.cpp \
    -L"$ROOT/dindel_tgi_amd/csrc" -ldindel_hmm -lz -pthread -Wl,-rpath,"$ROOT/dindel_tgi_amd/csrc"
cd "$ROOT"
# libstdc++ is preloaded too: python itself does not link it, and ASan's __cxa_throw interceptor must find the real one at start-up
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" ASAN_OPTIONS=detect_leaks=0 DD_ORACLE_LIB=/tmp/libdd_oracle_asan.so \
    DD_HOST_LIB=/tmp/libdindel_host_asan.so python -m pytest tests/test_oracle_kat.py tests/test_oracle_fast_cpu.py tests/test_host_adapter_cpu.py \
    tests/test_genotype_n1.py tests/test_cigar_cpu.py tests/test_n2_readers_cpu.py tests/test_glf_vcf_cpu.py tests/test_fast_inflate_cpu.py -q -m "not gpu" -p no:cacheprovider

# ThreadSanitizer over the window loop's threads (reader, prepare workers, ordered hand-overs, recycled batches, the haplotype file
# parsed on demand): the driver in --prepareOnly mode needs no GPU.  (The compute and reduce stages proper run on the GPU box only.)
H="$ROOT/dindel_tgi_amd/host"
g++ -O1 -g -std=c++11 -pthread -fsanitize=thread -I"$H" -o /tmp/dindel_gpu_tsan "$H"/dindel_gpu.cpp \
    "$H"/{compute_likelihoods,genotype,cigar,glf_to_vcf,bam_reader,window_io,get_reads,diploid_glf,realigned_bam,fast_inflate}.cpp \
    -L"$ROOT/dindel_tgi_amd/csrc" -ldindel_hmm -lz -Wl,-rpath,"$ROOT/dindel_tgi_amd/csrc"
python tools/n2_pipeline_bench.py --windows 600 --dir /tmp/tsan_sample --procs 4 > /dev/null 2>&1 || true      # writes the sample; its own driver run needs a GPU
TSAN_OPTIONS="halt_on_error=1" /tmp/dindel_gpu_tsan --bamFile /tmp/tsan_sample/reads.bam --varFile /tmp/tsan_sample/windows.txt \
    --hapFile /tmp/tsan_sample/haps.txt --outputFile /tmp/tsan_sample/o --prepareOnly --quiet --batchWindows 32 --prepareThreads 4
# the same build over two BAM pools with injected late skips: the writer's re-preparation (own handles, own read buffer) beside the prepare workers
python - <<'PY'
import sys, pathlib
sys.path.insert(0, ".")
from tests.test_n2_pools_cpu import _scene
p = pathlib.Path("/tmp/tsan_pools"); p.mkdir(exist_ok=True)
_scene(p, n_ref=40000, n_reads=8000)
PY
TSAN_OPTIONS="halt_on_error=1" /tmp/dindel_gpu_tsan --bamFiles /tmp/tsan_pools/bams.txt --varFile /tmp/tsan_pools/windows.txt --hapFile /tmp/tsan_pools/haps.txt \
    --outputFile /tmp/tsan_pools/o --prepareOnly --quiet --batchWindows 3 --prepareThreads 4 --injectLateSkip 4,9,10,20 2> /dev/null
echo "thread sanitizer: clean"
