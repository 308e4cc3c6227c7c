#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; : > $O/h_fail.txt
for i in $(seq 1 14); do
  timeout 200 python -m pytest tests/test_kernels_gpu.py -q -x -k "ff_block" > /tmp/h_$i.txt 2>&1
  if grep -q failed /tmp/h_$i.txt; then echo "== run $i" >> $O/h_fail.txt; grep -v "^$" /tmp/h_$i.txt | cut -c1-700 | tail -60 >> $O/h_fail.txt; fi
  tail -1 /tmp/h_$i.txt
done
