#!/bin/bash
# SQ-side PMC passes over the default bench (3 steps): where the waves of each kernel spend their cycles.
# SQ counters only (TA_/TCP_ counter passes hung a box in round 1).  Output: gpurun_out/pmc_sq{1,2,3}/
set -e
cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
P2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL"
P3="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU"
i=1
for P in "$P1" "$P2" "$P3"; do
  rocprofv3 --kernel-trace --pmc $P -d gpurun_out/pmc_sq$i -o sq --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --no-events --steps 3 --warmup 1 > gpurun_out/pmc_sq$i.log 2>&1
  i=$((i+1))
done
python profiles/tools/sq_summary.py gpurun_out/pmc_sq1/sq_counter_collection.csv gpurun_out/pmc_sq2/sq_counter_collection.csv gpurun_out/pmc_sq3/sq_counter_collection.csv
