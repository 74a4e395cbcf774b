#!/bin/bash
# an in-order eager run of the C1-sized probabilistic step with NaN-poisoned allocations: NaNs in the result = a kernel read memory nobody wrote
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "M1_PQ_LANES=0 M1_STREAMS=0" "M1_STREAMS=1"; do
  env M1_DEBUG_POISON=1 $cfg M1_BENCH_DUMP=/tmp/poison.pt python bench.py --workload ${WL:-C1P} --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-secondary 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
  python - <<PY
import torch
d = torch.load("/tmp/poison.pt")
off = 0
bad = []
for name, n in d["layout"]:
    g = d["grad"][off:off + n]
    if not torch.isfinite(g).all(): bad.append((name, int((~torch.isfinite(g)).sum()), n))
    off += n
print("$cfg: non-finite gradient entries in", len(bad), "parameters", bad[:12])
PY
done
