#!/bin/bash
R=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/pmc_a1 -o a1 -- python $R/tools/profile_step.py --steps 2 > $R/gpurun_out/pmc_a1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d $R/gpurun_out/pmc_a2 -o a2 -- python $R/tools/profile_step.py --steps 2 > $R/gpurun_out/pmc_a2.log 2>&1
cd $R
for d in pmc_a1 pmc_a2; do db=$(find gpurun_out/$d -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_summary.py $db | grep -E "k_attn_out_glu|k_conv_ffn<15, false, true>" | grep -v "^void ppasr::k.*| [0-9]* | [0-9.]* |" ; done
tail -3 gpurun_out/pmc_a2.log
