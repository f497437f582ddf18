#!/bin/bash
# MFMA-busy / wait-state counters of the shipped kernels (run on the GPU box through gpurun; TAG = e.g. r02a).
# Separate --pmc passes (8 SQ slots each), kernel trace only, the program directly after `--`, side stream off so each
# kernel has the chip to itself.  tools/pmc_mfma_summary.py turns the CSVs into profiles/<TAG>_pmc_mfma.json.
set -u
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export VDQN_NO_OVERLAP=1
ARGS="--steps 2 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-profile --pool 1"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE \
  -d $O/p1 -o p --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA \
  -d $O/p2 -o p --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $O/p2.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA \
  -d $O/p3 -o p --output-format csv -- python3 $R/bench.py $ARGS > /dev/null 2> $O/p3.err
find $O -name "*counter_collection.csv" | head
python3 $R/tools/pmc_mfma_summary.py $O $R/gpurun_out/${TAG}_pmc_mfma.json
