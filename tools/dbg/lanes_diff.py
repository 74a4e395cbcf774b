"""Which parameters end differently between the lane run and the in-order run of the RCCL harness configuration (C1P, world of one):
python tools/dbg/lanes_diff.py [runs] [steps] [extra VAR=v for the lane run ...]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = sys.argv[2] if len(sys.argv) > 2 else "1"
extra = dict(kv.split("=", 1) for kv in sys.argv[3:])
tmp = tempfile.mkdtemp()
def run(tag, env_extra, port):
    out = os.path.join(tmp, tag + ".pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", M1_BENCH_DUMP=out, **env_extra)
    if not os.environ.get("NODIST") and not (tag == "inorder" and os.environ.get("REF_NODIST")): env["M1_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "C1P", "--steps", steps, "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline"] + (["--no-graph"] if os.environ.get("NOGRAPH") else []), env=env, capture_output=True, text=True)
    if r.returncode:
        print(tag, "rc", r.returncode, r.stderr[-600:]); return None
    return torch.load(out, weights_only=False)
ref = run("inorder", {"M1_PQ_LANES": "0", "M1_STREAMS": "0"}, 29700)
for i in range(runs):
    ex_i = dict(extra)
    if os.environ.get("PADRAND"):                                     # a different allocation layout in every process
        import random
        ex_i["M1_BENCH_PAD_MB"] = ",".join(str(round(random.uniform(0.1, 700.0), 3)) for _ in range(3))
    d = run(f"lanes{i}", ex_i, 29701 + i)
    if d is None: continue
    for k_, ents in d.get("dbg", {}).items():
        for j, (ea, eb) in enumerate(zip(ents, ref["dbg"][k_])):
            print(f"run {i}: dbg {k_}[{j}]:", ["same" if torch.equal(x, y) else f"DIFF max {float((x.float() - y.float()).abs().max()):.3g} of {float(y.float().abs().max()):.3g}" for x, y in zip(ea, eb)])
    print(f"run {i}: step {d['step'].tolist()} (ref {ref['step'].tolist()})  rng {d['rng'].tolist()} (ref {ref['rng'].tolist()})")
    for key in ("grad", "flat"):
        a, b = d[key], ref[key]
        if torch.equal(a, b): print(f"run {i}: {key} identical"); continue
        off, rows, same = 0, [], []
        for name, n in ref["layout"]:
            nd = int((a[off:off + n] != b[off:off + n]).sum())
            if nd: rows.append((name, n, nd, float((a[off:off + n] - b[off:off + n]).abs().max()), float(b[off:off + n].abs().max())))
            else: same.append(name)
            off += n
        print(f"run {i}: {key}: {len(rows)} parameters differ of {len(ref['layout'])}")
        if key == "grad" and os.environ.get("VALS"):
            off2 = 0
            for name, n in ref["layout"]:
                if name.endswith(os.environ["VALS"]):
                    print("      ", name, "lane:", [round(float(v), 3) for v in a[off2:off2 + 10]], "\n         ref:", [round(float(v), 3) for v in b[off2:off2 + 10]])
                off2 += n
        for r in rows[:int(os.environ.get('ROWS', '8'))]: print("      ", r)
        import collections
        kinds = collections.Counter((n_.split('.')[1], n_.split('.')[-1], 'diff') for n_, *_ in rows) + collections.Counter((n_.split('.')[1], n_.split('.')[-1], 'same') for n_ in same)
        print('       by (net, kind, state):', sorted(kinds.items()))
        if os.environ.get('SAME'): print('       same:', same)
