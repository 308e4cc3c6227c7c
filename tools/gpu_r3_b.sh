#!/bin/bash
# round-3 GPU call B: fused FF kernel - where does the time go (ablations, DMA order, PMC)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "ff_block_matches" 2>&1 | tail -12 > $O/b_tests.txt
cat $O/b_tests.txt
timeout 300 python tools/ff_probe.py > $O/b_ff_probe.txt 2>&1
cat $O/b_ff_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/b_pmc1 -o p -- python $R/tools/ff_pmc.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b_pmc2 -o p -- python $R/tools/ff_pmc.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/b_pmc3 -o p -- python $R/tools/ff_pmc.py > /dev/null 2>&1
cd $R
for d in b_pmc1 b_pmc2 b_pmc3; do python tools/pmc_summary.py $O/$d ff_block; done > $O/b_pmc_summary.txt 2>&1
cat $O/b_pmc_summary.txt
find $O/b_pmc1 $O/b_pmc2 $O/b_pmc3 -name "*.csv" -size +2M -delete
