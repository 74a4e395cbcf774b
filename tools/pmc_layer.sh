#!/bin/bash
# usage (on the GPU box): bash tools/pmc_layer.sh <layer> [outdir]   -- separate --pmc passes over tools/bench_conv.py <layer>
R=${GRAFT_REPO_ROOT:-$(pwd)}; L=$1; O=$R/${2:-gpurun_out/pmc_$L}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i --output-format csv -- python3 $R/tools/bench_conv.py $L > $O/p$i.out 2> $O/p$i.err
done
cd $R; python tools/pmc_agg.py $O conv_mfma_kernel
