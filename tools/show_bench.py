"""Short human-readable view of a bench.py JSON line."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])


def show(tag, d):
    r = d.get("roofline") or {}
    print(f"{tag}: {d['value']:.1f} {d['unit']}  {d['ms_per_step']:.2f} ms/step  graph={d['config']['hip_graph']} err={d['config']['graph_error']}")
    if r:
        iso = r.get("isolated", {})
        print(f"   dominant {r['kernel']} {r['bound']} frac {r['frac']:.4f} (isolated {iso.get('frac', 0):.4f}) {r['kernel_ms_per_step']:.2f} ms/step traffic {r['traffic']}")
        print("   " + "  ".join(f"{k} {v:.2f}" for k, v in r["all_kernels_ms_per_step"].items()))


show("headline", d)
if d.get("cpu_baseline"):
    print("   cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][-80:])
for k, v in (d.get("secondary") or {}).items():
    show(k, v)
