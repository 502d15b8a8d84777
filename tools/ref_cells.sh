#!/bin/bash
# GPU box: the reference's own round-trip test (build/ref_tests_shard) over every (distribution, bytesoftype) cell, a bounded
# time each; one line per cell: round trips done, failures.  usage: bash tools/ref_cells.sh [seconds per cell] > gpurun_out/ref_cells.txt
SEC=${1:-15}
for d in same sorted random; do
  for T in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
    timeout -k 5 $SEC build/ref_tests_shard $d $T > /tmp/ref_cell.log 2>&1; rc=$?
    echo "$d $T rc=$rc done=$(grep -c done /tmp/ref_cell.log) errors=$(grep -c 'Test error' /tmp/ref_cell.log) last=$(tail -1 /tmp/ref_cell.log | cut -c1-80)"
  done
done
