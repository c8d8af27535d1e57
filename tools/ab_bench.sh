#!/bin/bash
# A/B of the headline bench line between the shipped library and a tuning variant, alternating on one box:
#   bash tools/ab_bench.sh VARIANT [rounds]
R=$(cd "$(dirname "$0")/.." && pwd)
V=$R/pathtracer_amd/lib/variants/libptx_$1.so
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step; occlusion', d.get('anyhit_mrays_per_s'), 'reference order', d.get('reference_mode_mrays_per_s'))"; }
for i in $(seq ${2:-2}); do
  timeout -k 10 200 python3 $R/bench.py --no-render --no-cpu --steps 20 2>/dev/null | show shipped || exit 1
  PTX_AMD_LIB=$V timeout -k 10 200 python3 $R/bench.py --no-render --no-cpu --steps 20 2>/dev/null | show $1 || exit 1
done
