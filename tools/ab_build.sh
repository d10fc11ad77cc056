#!/bin/bash
# A/B build of the library with ONE (K, D) instantiation of the main kernel and extra -D switches (seconds instead of minutes):
#   tools/ab_build.sh NAME K D [extra hipcc flags...]   ->  tools/_ab/lib_NAME.so   (git-ignored; travels to the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/dindel_tgi_amd/csrc
name=$1; K=$2; D=$3; shift 3
mkdir -p $R/tools/_ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-result -DDD_ONLY_K=$K -DDD_ONLY_D=$D "$@" -I$SRC \
    -shared -o $R/tools/_ab/lib_$name.so $SRC/hmm_kernel.hip $SRC/genotype_kernel.hip $SRC/faster_kernel.hip $SRC/capi.cpp
echo built tools/_ab/lib_$name.so
