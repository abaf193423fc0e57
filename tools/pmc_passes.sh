#!/bin/bash
# GPU box: SQ counter passes over the isolated hot-kernel launches (tools/pmc_probe.py), folded per kernel.
# usage: tools/pmc_passes.sh [kernel-name filter]      (one rocprofv3 --pmc run per counter group)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAVES_EQ_64"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmcq_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcq_$i -- python3 $R/tools/pmc_probe.py 2048 > $R/gpurun_out/pmcq_$i.log 2>&1 || echo "group $i failed: $grp"
  python3 $R/tools/pmc_fold.py $R/gpurun_out/pmcq_$i "${1:-k_cycle0<4, 8, 8, true, false, false, 19>}"
done
