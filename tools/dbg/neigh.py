"""kernels around / overlapping the launches of one (kernel substring, grid.x): python tools/dbg/neigh.py trace.csv substr gridx [max]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); sub = sys.argv[2]; gx = int(sys.argv[3]); mx = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ks = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'^void ', '', r['Kernel_Name'])[:70],
              (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])), r.get('Queue_Id', '?'),
              int(r['Workgroup_Size_X']), r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', '?'), r.get('Scratch_Size', '?')) for r in rows))
t0 = ks[0][0]; shown = 0
for i, k in enumerate(ks):
    if sub in k[2] and k[3][0] == gx:
        print(f"--- #{i} {k[2]} grid={k[3]} wg={k[5]} lds={k[6]} vgpr={k[7]} scratch={k[8]} q={k[4]} start {(k[0]-t0)/1e3:.1f} us dur {(k[1]-k[0])/1e3:.1f} us")
        for j in range(max(0, i - 4), min(len(ks), i + 4)):
            o = ks[j]
            if j != i:
                ov = min(o[1], k[1]) - max(o[0], k[0])
                print(f"     {'OVERLAP' if ov > 0 else '       '} {(o[0]-k[0])/1e3:9.1f} .. {(o[1]-k[0])/1e3:9.1f} us q={o[4]} {o[2][:60]} grid={o[3]}")
        shown += 1
        if shown >= mx: break
