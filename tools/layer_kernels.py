"""Per-layer kernel time of the conv entry points inside an EAGER C3/C2 step: correlates the M1_MFMA_LOG lines (one per conv launch, in
launch order) with the conv kernels of a rocprofv3 kernel trace (in dispatch order).
usage: python tools/layer_kernels.py <kernel_trace.csv> <stderr log with 'mfma:' lines> [steps in the log/trace]"""
import collections, csv, re, sys
trace, log = sys.argv[1], sys.argv[2]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r['Dispatch_Id']))
ks = [(re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name'])), int(r['End_Timestamp']) - int(r['Start_Timestamp']),
       (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))) for r in rows]
convk = [k for k in ks if k[0].startswith(('conv_mfma_kernel', 'conv_halo_kernel', 'conv_pw_kernel', 'conv_t3_kernel'))]
lines = [l.strip() for l in open(log) if l.startswith('mfma:')]
print(f"{len(convk)} conv kernels in the trace, {len(lines)} log lines")
n = min(len(convk), len(lines))
agg = collections.defaultdict(lambda: [0, 0, None, None])
for (nm, dur, grid), l in zip(convk[-n:], lines[-n:]):
    key = re.sub(r'^mfma: ', '', l)
    a = agg[key]; a[0] += 1; a[1] += dur; a[2] = nm[:58]; a[3] = grid
tot = sum(a[1] for a in agg.values())
print(f"total {tot/1e6:.3f} ms over all logged launches")
for key, (c, t, nm, grid) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    m = re.search(r'N(\d+) out (\d+)x(\d+)x(\d+) CC (\d+) OC (\d+) k(\d+) taps s(\d)(\d)(\d)', key)
    N, D, H, W, CC, OC, taps = (int(m.group(i)) for i in range(1, 8))
    s = [int(m.group(i)) for i in (8, 9, 10)]
    mode = int(key.split()[1])
    vox = N * D * H * W
    macs = vox * taps * CC * OC / (s[0] * s[1] * s[2] if mode == 1 and taps > 1 and max(s) > 1 else 1)
    print(f"{t/1e3:9.1f} us n={c:3d} avg={t/c/1e3:7.1f} us {2*macs/(t/c)/1e3:7.1f} TF/s  grid={grid} {nm}\n            {key}")
