#!/bin/bash
# Output equality of the window loop across revisions of the host code: runs the current dindel_gpu and the drivers of earlier
# revisions kept under tools/_ab/<name>/ (git-ignored; built with
#   git archive <rev> dindel_tgi_amd/host include dindel_tgi_amd/csrc/hmm_kernel.h | tar -x -C /tmp/x && make -C /tmp/x/dindel_tgi_amd/host
# against the same libdindel_hmm.so) on the sample tools/n2_pipeline_bench.py left in $1, and compares the .glf.txt files byte for byte.
#   tools/n2_ab_compare.sh DIR            (after: python tools/n2_pipeline_bench.py --windows N --dir DIR)
set -u
D=$1
R=$(cd "$(dirname "$0")/.." && pwd)
TORCH_LIB=$(python3 -c 'import os, torch; print(os.path.join(os.path.dirname(torch.__file__), "lib"))' 2>/dev/null)
export LD_LIBRARY_PATH=$R/dindel_tgi_amd/csrc:$TORCH_LIB:/opt/rocm/lib:${LD_LIBRARY_PATH:-}
args="--bamFile $D/reads.bam --varFile $D/windows.txt --hapFile $D/haps.txt --quiet"
for model in "" "--faster"; do
    tag=main; [ -n "$model" ] && tag=faster
    s=$(date +%s.%N)
    "$R/dindel_tgi_amd/host/dindel_gpu" $args --outputFile "$D/cur_$tag" $model 2>/dev/null || { echo "current driver failed ($tag)"; exit 1; }
    e=$(date +%s.%N)
    echo "$tag current: $(python3 -c "print(round($e - $s, 3))") s, $(wc -l < "$D/cur_$tag.glf.txt") lines"
    for old in ${AB_DIRS:-"$R"/tools/_ab/*_host}; do        # AB_DIRS="dir ...": only these revisions
        [ -d "$old" ] || continue
        bin=$(ls "$old"/dindel_gpu_* 2>/dev/null | head -1)
        [ -n "$bin" ] || continue
        name=$(basename "$old")
        s=$(date +%s.%N)
        LD_LIBRARY_PATH=$old:$LD_LIBRARY_PATH "$bin" $args --outputFile "$D/${name}_$tag" $model 2>/dev/null || { echo "$name failed ($tag)"; exit 1; }
        e=$(date +%s.%N)
        if cmp -s "$D/cur_$tag.glf.txt" "$D/${name}_$tag.glf.txt"; then verdict="identical output"; else verdict="OUTPUT DIFFERS"; fi
        echo "$tag $name: $(python3 -c "print(round($e - $s, 3))") s, $verdict"
    done
done
