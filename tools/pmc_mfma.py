"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE pass per kernel:
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  (GRBM_GUI_ACTIVE is reported summed
over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES sums the matrix-pipe busy cycles of all 256 CUs x 4 SIMDs: 16 cycles per
v_mfma_f32_16x16x32_bf16, MI355X_MICROARCH.md "rocprofv3 PMC slots".)
usage: python tools/pmc_mfma.py <counter_collection.csv> <steps>"""
import collections, csv, re, sys
path, steps = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    k = re.sub(r"^void ", "", r["Kernel_Name"]); k = re.sub(r"\(.*", "", k)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
tot_busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in agg.values()); tot_act = sum(v["GRBM_GUI_ACTIVE"] for v in agg.values())
print(f"whole step ({steps} eager steps, {sum(cnt.values()) / steps:.0f} kernel launches per step): MFMA busy "
      f"{100 * tot_busy / (tot_act / 8 * 1024):.1f} % of GPU-active cycles; GPU-active {tot_act / 8 / steps / 1e6:.2f} Mcycles per step")
print(f"{'kernel':78s} {'n/step':>7s} {'Mcyc/step':>10s} {'MFMA busy %':>11s} {'mfma/launch':>12s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:28]:
    act = v["GRBM_GUI_ACTIVE"] / 8
    print(f"{k[:78]:78s} {cnt[k] / steps:7.1f} {act / steps / 1e6:10.3f} {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / max(act * 1024, 1):11.1f} "
          f"{v['SQ_INSTS_MFMA'] / max(cnt[k], 1):12.0f}")
