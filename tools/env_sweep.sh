#!/bin/bash
# One-variable experiments on the bench step: tools/env_sweep.sh VAR v1 v2 ...  (prints ms/step and the conv-family times)
VAR=$1; shift
cd /root/repo
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-other-configs --no-loss-check --steps 15 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels_ms_per_step']
print('$VAR=$v', 'ms/step %.2f' % d['ms_per_step'], 'fwd %.2f dgrad %.2f wgrad %.2f' % (k['conv_fwd_kernel'], k['conv_dgrad_kernel'], k['conv_wgrad_kernel']))"
done
