#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc_rr; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  for k in tblock ff; do
    timeout 300 rocprofv3 --pmc $set --output-format csv -d $O/${k}_$i -o p -- python $R/tools/${k}_pmc.py > /dev/null 2>&1
  done
done
python $R/tools/pmc_rr.py $O/* | tee $R/gpurun_out/l_pmc_rr.txt
rm -rf $O
