#!/bin/bash
# Dev aid (GPU box): bench line of config 3 reduced to the numbers watched while tuning
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python3 bench.py --no-cpu "$@" > gpurun_out/b.json || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/b.json').read().strip().splitlines()[-1])
k=lambda r: r['kernel'].split('(')[0].split(' ')[0][:24]
print('steps/s %.1f  ms/step %.3f  %s %.3f  %s %.3f  fwd %.3f s  bwd %.3f s' % (d['value'], d['ms_per_step'], k(d['roofline']), d['roofline']['avg_launch_ms'], k(d['roofline_second_kernel']), d['roofline_second_kernel']['avg_launch_ms'], d['config']['forward_s'], d['config']['backward_s']))"
