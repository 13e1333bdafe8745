#!/bin/bash
# Same-box comparison of builds of the library that differ in V1 only (tools/build/ab/lib*.so): tools/v1_bubble_probe.py with
# each build in turn, three rounds, so that box-to-box and warm-up differences cancel.
mkdir -p gpurun_out/r04
out=gpurun_out/r04/v1_ab.txt
: > $out
for round in 1 2 3; do
  for v in "$@"; do
    echo "== $v round $round" >> $out
    HITADV_LIBRARY=$PWD/tools/build/ab/lib$v.so python tools/v1_bubble_probe.py 2>/dev/null | tr -d '\n ' >> $out
    echo >> $out
  done
done
cat $out
