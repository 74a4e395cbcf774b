#!/bin/bash
# Lists every kernel of csrc/ that uses scratch (private) memory or spills VGPRs.  A run-time index into a by-value kernel argument
# struct moves the WHOLE struct to scratch (round 2: p.Am[mem] in wgrad_tf cost 10 % of the C3 step); a run-time loop bound over a
# local vector does the same to that vector.  Expected output: only wgrad_tf64_kernel (experimental, off by default).
cd "$(dirname "$0")/../prostatemr_3d-cad-cspca_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc -S --offload-arch=gfx950 -O3 --cuda-device-only -o /tmp/scratch_$f.s $f 2>/dev/null
  grep "private_segment_fixed_size:\|\.name:\|vgpr_spill_count:" /tmp/scratch_$f.s | paste - - - | awk -v f=$f '($4+0 > 0 || $6+0 > 0) {print f, $2, "scratch", $4, "vgpr spills", $6}' | cut -c1-160
  rm -f /tmp/scratch_$f.s
done
