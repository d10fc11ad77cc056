#!/bin/bash
# A/B of main-kernel builds: tools/ab_main.sh lib1.so lib2.so ...   (run on the GPU box)
for lib in "$@"; do
  DD_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --windows ${WINDOWS:-10000} --steps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],1), 'ms', '%.4g' % d['value'], 'cells/s')"
done
