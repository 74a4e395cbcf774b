#!/bin/bash
# usage (on the GPU box): bash tools/pmc_layer.sh <layer> [kernel-substring] [outdir]  -- separate --pmc passes over tools/bench_conv.py <layer>
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=$1; K=${2:-conv_mfma_kernel}; O=$R/${3:-gpurun_out/pmc_$L}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i --output-format csv -- python3 $R/tools/bench_conv.py $L > $O/p$i.out 2> $O/p$i.err
done
cd $R; python tools/pmc_agg.py $O $K
