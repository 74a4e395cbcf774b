#!/bin/bash
# usage (on the GPU box): bash tools/pmc_cmd.sh <outdir> <kernel-substring> <python script + args ...>
# separate rocprofv3 --pmc passes (SQ timing / instruction mix / LDS) over one python command; prints mean counters per launch
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/$1; K=$2; shift 2
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i --output-format csv -- python3 $R/"$1" "${@:2}" > $O/p$i.out 2> $O/p$i.err
done
cd $R; python tools/pmc_agg.py $O $K
