"""Which step, and which op of that step, first differs between a reference process and N processes of a second configuration
(C1P harness workload; default: reference = no process group, others = RCCL process group present, no collective issued):

    python tools/dbg/first_diff.py N [REFVAR=v ...] -- [VAR=v ...]

Every process runs bench.py with M1_BENCH_HIST=1 (hash of the gradient / parameter vectors after every executed step) and
M1_DEBUG_TRACE (per-op output checksums, also inside the replayed graph), everything in order on one stream unless overridden.
Prints per process: the end state (short hash: how many distinct states exist), the first step whose hashes differ from the
reference and the first op of that step whose output checksum differs."""
import collections, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rest = sys.argv[2:]
cut = rest.index("--") if "--" in rest else len(rest)
ref_env = dict(kv.split("=", 1) for kv in rest[:cut])
run_env = dict(kv.split("=", 1) for kv in rest[cut + 1:])
tmp = tempfile.mkdtemp()
BASE = {"M1_PQ_LANES": "0", "M1_STREAMS": "0", "M1_BENCH_HIST": "1", "M1_DEBUG_TRACE": "16384", "M1_DEBUG_FIXED_EPS": "1"}
PG = {"M1_BENCH_FORCE_DIST": "1", "M1_BENCH_NO_COLLECTIVES": "1"}
steps = os.environ.get("STEPS", "2")


def run(tag, extra, port):
    out = os.path.join(tmp, tag + ".pt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", M1_BENCH_DUMP=out)
    env.update(BASE); env.update(extra)
    for k in [k for k, v in env.items() if v == "-"]:
        del env[k]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "C1P", "--steps", steps, "--warmup", "1",
                        "--no-cpu-baseline", "--no-roofline"] + (["--no-graph"] if env.get("NOGRAPH") else []), env=env, capture_output=True, text=True)
    if r.returncode:
        print(tag, "rc", r.returncode, r.stderr[-800:]); return None
    return torch.load(out, weights_only=False)


ref = run("ref", ref_env, 29800)
assert ref is not None
print("reference:", " ".join(f"{h['tag']}:{h['grad']}" for h in ref["hist"]))
ends = collections.Counter()
for i in range(n):
    d = run(f"run{i}", dict(PG, **run_env), 29801 + i)
    if d is None:
        continue
    end = d["hist"][-1]["flat"]
    ends[end] += 1
    first = next((j for j, (a, b) in enumerate(zip(d["hist"], ref["hist"])) if a["grad"] != b["grad"] or a["flat"] != b["flat"]), None)
    if first is None:
        print(f"run {i}: end {end} == reference in every step"); continue
    a, b = d["hist"][first], ref["hist"][first]
    msg = f"run {i}: end {end}; first differing step #{first} ({a['tag']}) grad {a['grad']} vs {b['grad']}"
    ta, tb = a.get("trace"), b.get("trace")
    if ta and tb:
        (na, ca), (nb, cb) = ta, tb
        if len(na) != len(nb):
            msg += f"; op lists differ in length ({len(na)} vs {len(nb)})"
        k = next((k for k in range(min(len(ca), len(cb))) if int(ca[k]) != int(cb[k]) or na[k] != nb[k]), None)
        if k is None:
            msg += "; no op output differs (the difference is in an accumulated buffer)"
        else:
            ndiff = sum(int(ca[q]) != int(cb[q]) for q in range(min(len(ca), len(cb))))
            msg += f"; first differing op output: slot {k} of {len(ca)}: {na[k]} (previous: {na[k - 1] if k else None}); {ndiff} slots differ"
            if os.environ.get("LIST"):
                for q in range(k, min(len(ca), k + int(os.environ["LIST"]))):
                    print("      ", q, "DIFF" if int(ca[q]) != int(cb[q]) else "same", na[q])
    print(msg)
print("end states:", dict(ends), "reference end:", ref["hist"][-1]["flat"])
