#!/bin/bash
# what each family of entry points costs the REPLAYED step: M1_DEBUG_SKIP=<family> makes its entry points return without launching
# (results are garbage); one bench line per family.  usage (GPU box): bash tools/dbg/skip_families.sh [WL]
cd "$(dirname "$0")/../.."
export WL=${1:-C3}
bash tools/sweep_c3.sh "M1_DEBUG_SKIP=conv_wgrad" "M1_DEBUG_SKIP=conv_wgrad,fold" "M1_DEBUG_SKIP=fold" "M1_DEBUG_SKIP=conv_dgrad" "M1_DEBUG_SKIP=conv_fwd" \
  "M1_DEBUG_SKIP=in_apply,in_bwd" "M1_DEBUG_SKIP=se_fwd,se_bwd" "M1_DEBUG_SKIP=gate" "M1_DEBUG_SKIP=adam" "M1_DEBUG_SKIP=pack" \
  "M1_DEBUG_SKIP=conv_wgrad,fold,conv_dgrad,conv_fwd" "M1_DEBUG_SKIP=in_apply,in_bwd,se_fwd,se_bwd,gate,adam,pack"
