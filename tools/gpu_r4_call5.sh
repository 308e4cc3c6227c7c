#!/bin/bash
# round 4, GPU call 5: the whole -m gpu suite on the current library (the packed epilogue had dropped the batch offset: VAE
# attention), ring depth of the 128x64 tile on the latency-bound small-M linears
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
PROBE_SWEEP=1 PROBE_SMALL=1 PROBE_CFGS=0,2,770,1026,1,769,6 timeout 400 python tools/gemm_probe.py > $O/c5_probe_small.txt 2>&1; tail -12 $O/c5_probe_small.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $O/c5_gpu_suite.txt 2>&1; tail -6 $O/c5_gpu_suite.txt
cp gpurun_out/parity_report.txt $O/c5_parity_report.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/c5_smoke.txt 2>&1; tail -2 $O/c5_smoke.txt
