#!/bin/bash
# In-box A/B of an environment knob: tools/ab_env.sh "<bench args>" VAR v1 v2 ...   (three interleaved rounds on one box)
args="$1"; var="$2"; shift 2
for round in 1 2 3; do
  for v in "$@"; do
    env $var=$v python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round $var=$v: %.3f graphs/s' % d['value'])"
  done
done
