#!/bin/bash
# A/B harness for kernel variants: build each variant here (VDQN_EXTRA_FLAGS=... python -m video_dqn_amd.build --force; cp lib/libvdqn.so
# lib/libvdqn_<name>.so), then on ONE gpu box: bash tools/ab_libs.sh <name> <name> ...  (alternates the variants, 3 rounds)
L=video_dqn_amd/lib
cp $L/libvdqn.so /tmp/libvdqn_keep.so
for round in 1 2 3; do
  for v in "$@"; do
    cp $L/libvdqn_$v.so $L/libvdqn.so
    python bench.py --steps 150 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['achieved'])"
  done
done
cp /tmp/libvdqn_keep.so $L/libvdqn.so
