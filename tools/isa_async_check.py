#!/usr/bin/env python3
"""Static check of the compiler output for registers that are touched while an asynchronous load issued FROM INLINE ASM is still
in flight.

hipcc's waitcnt insertion does not see loads issued inside ``asm volatile`` (``ds_read_b128``, ``ds_read_b64_tr_b16``,
``global_load_dwordx2`` ... in conv_mfma / conv_halo / conv_t3 / wgrad_tap / wgrad_tf): the kernels wait for them by hand
(``s_waitcnt`` statements tied to the fragments).  The register allocator is still free to COPY such an asm output (v_mov, a
spill, a tuple re-pack) between the load and the hand-written wait -- the copy then reads whatever the register held before the
data landed: a run-dependent value that looks like data (an earlier fragment), the signature of the state-dependent results of
rounds 4/5.  This tool walks the ``-S`` output of every kernel and reports each instruction that reads or writes a destination
register of an asm-issued load before a wait that covers it.

    python tools/isa_async_check.py [file.hip ...]       (default: every csrc/*.hip, production flags of csrc/Makefile)

Every kernel's basic blocks are walked along the control-flow graph (path-sensitive: the queue of outstanding loads is part of the
state, states are memoised per block).  Exit status 1 when anything is reported.
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "prostatemr_3d-cad-cspca_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        k = m.group(1)
        if m.group(2) is not None:
            out.add((k, int(m.group(2))))
        else:
            out.update((k, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def makefile_flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    nopk = re.search(r"^NOPK\s*:?\??=\s*(.*)$", mk, re.M).group(1).split()
    return ["-O3", "-std=c++17", "--offload-arch=gfx950"] + nopk


def compile_s(src, flags):
    out = f"/tmp/isa_check_{os.path.basename(src)}.s"
    subprocess.run([HIPCC, "-S", "--cuda-device-only", *flags, "-o", out, os.path.abspath(src)], check=True, cwd=CSRC, stderr=subprocess.DEVNULL)
    return out


LGKM_OPS = ("ds_", "s_load", "s_buffer_load", "s_sendmsg", "s_memtime", "s_memrealtime")
VM_OPS = ("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic", "flat_load", "flat_store",
          "flat_atomic", "scratch_load", "scratch_store")


def parse_wait(ins):
    """-> (vmcnt or None, lgkmcnt or None) of an s_waitcnt."""
    vm = lg = None
    m = re.search(r"vmcnt\((\d+)\)", ins)
    if m:
        vm = int(m.group(1))
    m = re.search(r"lgkmcnt\((\d+)\)", ins)
    if m:
        lg = int(m.group(1))
    if vm is None and lg is None:
        m = re.search(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", ins)
        if m:                                           # raw immediate: gfx9 encoding vmcnt[3:0]|[15:14], expcnt[6:4], lgkmcnt[11:8]
            v = int(m.group(1), 0)
            vm = (v & 0xF) | ((v >> 14) & 0x3) << 4
            lg = (v >> 8) & 0xF
    return vm, lg


def split_kernels(path):
    """-> {kernel name: [(line no, instruction text, in_asm)] + labels as ('label', name)}"""
    kernels, cur, in_asm = {}, None, False
    for no, raw in enumerate(open(path).read().splitlines(), 1):
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^([A-Za-z_$][\w$.]*):", s)
        if m:
            name = m.group(1)
            if name.startswith(".L") or name.startswith("L"):
                if cur is not None:
                    cur.append((no, "label", name))
            elif not name.startswith("."):
                cur = kernels.setdefault(name, [])
            continue
        if cur is None or not s or s.startswith(";") or s.startswith("."):
            continue
        ins = s.split(";")[0].strip()
        if ins:
            cur.append((no, ins, in_asm))
    return kernels


def build_blocks(items):
    """Basic blocks of one kernel: list of dicts {label, ins: [(no, text, in_asm)], succ: [block index]}."""
    blocks, cur = [], {"label": None, "ins": []}
    for it in items:
        if it[1] == "label":
            if cur["ins"] or cur["label"] is not None:
                blocks.append(cur)
            cur = {"label": it[2], "ins": []}
            continue
        cur["ins"].append(it)
        op = it[1].split()[0]
        if op.startswith("s_branch") or op.startswith("s_cbranch") or op in ("s_endpgm", "s_setpc_b64"):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"] or cur["label"] is not None:
        blocks.append(cur)
    by_label = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    for i, b in enumerate(blocks):
        succ = []
        last = b["ins"][-1][1] if b["ins"] else ""
        op = last.split()[0] if last else ""
        if op.startswith("s_branch"):
            t = last.split()[-1]
            if t in by_label:
                succ.append(by_label[t])
        elif op.startswith("s_cbranch"):
            t = last.split()[-1]
            if t in by_label:
                succ.append(by_label[t])
            if i + 1 < len(blocks):
                succ.append(i + 1)
        elif op in ("s_endpgm", "s_setpc_b64"):
            pass
        elif i + 1 < len(blocks):
            succ.append(i + 1)
        b["succ"] = succ
    return blocks


def run_block(b, lg_q, vm_q, findings, kernel):
    lg_q, vm_q = list(lg_q), list(vm_q)
    for no, ins, in_asm in b["ins"]:
        op = ins.split()[0]
        if op == "s_waitcnt":
            vm, lg = parse_wait(ins)
            if lg is not None and len(lg_q) > lg:
                lg_q = lg_q[len(lg_q) - lg:] if lg else []
            if vm is not None and len(vm_q) > vm:
                vm_q = vm_q[len(vm_q) - vm:] if vm else []
            continue
        if op == "s_endpgm":
            return [], []
        touched = regs_of(ins)
        for q, qname in ((lg_q, "lgkm"), (vm_q, "vm")):
            for dst, lno, txt in q:
                if not dst:
                    continue
                hit = touched & dst
                if hit:
                    same_queue_load = (qname == "lgkm" and op.startswith("ds_read")) or (qname == "vm" and op.startswith(VM_OPS) and "load" in op)
                    if same_queue_load and in_asm:
                        # a later asm load of the same in-order queue may re-target an in-flight destination (returns in order);
                        # its ADDRESS operand naming an in-flight destination is a stale read
                        addr_regs = regs_of(",".join(ins[len(op):].split(",")[1:]))
                        if not (addr_regs & dst):
                            continue
                    findings.add((kernel, no, ins, lno, txt, tuple(sorted(hit)[:4])))
        if op.startswith(LGKM_OPS):
            dst = frozenset()
            if in_asm and op.startswith("ds_read"):
                dst = frozenset(regs_of(ins[len(op):].split(",")[0]))
            lg_q.append((dst, no, ins))
        elif op.startswith(VM_OPS):
            dst = frozenset()
            if in_asm and "load" in op and not re.search(r"\blds\b", ins):
                dst = frozenset(regs_of(ins[len(op):].split(",")[0]))
            vm_q.append((dst, no, ins))
    return lg_q, vm_q


def trim(q, cap=80):
    """Only asm-load entries matter; entries older than the oldest asm load carry no information."""
    first = next((i for i, e in enumerate(q) if e[0]), None)
    if first is None:
        return ()
    q = q[first:]
    return tuple(q[-cap:])


def check_file(path, max_visits=400000):
    findings = set()
    for kernel, items in split_kernels(path).items():
        if not any(it[1] != "label" and it[2] for it in items):
            continue                                      # no inline asm in this function
        # inline asm that writes M0 (LDS-DMA base) does not declare it: no compiler-generated instruction of the same kernel may use M0
        if any(it[1] != "label" and it[2] and re.search(r"\bm0\b", it[1]) for it in items):
            for it in items:
                if it[1] != "label" and not it[2] and re.search(r"\bm0\b", it[1]):
                    findings.add((kernel, it[0], it[1], 0, "inline asm of this kernel writes m0 undeclared", (("m", 0),)))
        blocks = build_blocks(items)
        if not blocks:
            continue
        seen = set()
        work = [(0, (), ())]
        visits = 0
        while work:
            bi, lg, vm = work.pop()
            key = (bi, tuple((e[0], e[1]) for e in lg), tuple((e[0], e[1]) for e in vm))
            if key in seen:
                continue
            seen.add(key)
            visits += 1
            if visits > max_visits:
                print(f"   ({kernel[:60]}: state cap reached, result partial)")
                break
            lg2, vm2 = run_block(blocks[bi], lg, vm, findings, kernel)
            lg2, vm2 = trim(lg2), trim(vm2)
            for s_ in blocks[bi]["succ"]:
                work.append((s_, lg2, vm2))
    return sorted(findings, key=lambda f: (f[0], f[1]))


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    flags = makefile_flags()
    bad = 0
    for src in srcs:
        spath = src if src.endswith(".s") else compile_s(src, flags)
        f = check_file(spath)
        print(f"{os.path.basename(src)}: {len(f)} finding(s)")
        for kernel, no, ins, lno, txt, hit in f[:40]:
            print(f"   {kernel[:70]}  line {no}: `{ins}` touches {hit} of in-flight `{txt}` (line {lno})")
        bad += len(f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
