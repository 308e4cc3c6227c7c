#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/j_gputests.txt; cat $O/j_gputests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/j_smoke.txt
