#!/bin/bash
# builds and runs tools/host_adapter_bench.cpp on the GPU box: tools/host_adapter_bench.sh [windows] [faster 0/1]
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
g++ -O2 -std=c++11 -I$R/dindel_tgi_amd/host -I$R/include $R/tools/host_adapter_bench.cpp -L$R/dindel_tgi_amd/host -ldindel_host \
    -L$R/dindel_tgi_amd/csrc -ldindel_hmm -Wl,-rpath,$R/dindel_tgi_amd/host -Wl,-rpath,$R/dindel_tgi_amd/csrc -o /tmp/hab
# torch ships the HIP runtime the library was linked against
export LD_LIBRARY_PATH=$(python3 -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))"):/opt/rocm/lib:$LD_LIBRARY_PATH
/tmp/hab ${1:-500} ${2:-0}
