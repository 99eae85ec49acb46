#!/bin/bash
# In-box A/B of library variants: tools/ab_libs.sh "<bench args>" A B C ...   (variants = tools/bin/ab/libdsg_<X>.so)
# Box-to-box differences (±1 %) are larger than most kernel tweaks; variants are therefore interleaved on one box.
args="$1"; shift
cp diffusesg_amd/lib/libdsg.so /tmp/libdsg_keep.so
for round in 1 2 3; do
  for v in "$@"; do
    cp tools/bin/ab/libdsg_$v.so diffusesg_amd/lib/libdsg.so
    python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round variant $v: %.3f graphs/s  gemm %.1f TF' % (d['value'], d['roofline']['achieved']))"
  done
done
cp /tmp/libdsg_keep.so diffusesg_amd/lib/libdsg.so
