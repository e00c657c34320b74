#!/bin/bash
# Dev tool: A/B of an environment switch on the benchmark's step, alternating runs in one gpurun call.  tools/ab_env.sh VAR a b [reps]
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do
  for v in $A $B; do
    echo -n "$VAR=$v  "
    env $VAR=$v python bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], 'img/s')"
  done
done
