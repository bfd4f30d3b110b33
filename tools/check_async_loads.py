#!/usr/bin/env python3
"""Lint for the hand-counted asynchronous loads of the HIP kernels (conv_h2.hip issues its halo loads as inline-asm
global_load_dwordx4 and waits for them with counted s_waitcnt vmcnt(N) much later; the compiler believes the destination registers
are written AT the asm statement).

Walks the gfx950 disassembly of every kernel in a code object (all branch targets, in-flight state carried along) with the
hardware's vmcnt model - vector-memory instructions retire in issue order, s_waitcnt vmcnt(N) leaves the N youngest in flight -
and reports every instruction that reads or writes a VGPR a load is still in flight to, and every s_endpgm reached with an
LDS-DMA (global_load_lds) still in flight (the block's LDS may belong to another block by the time it lands). hipcc's own loads never show up (it
waits before it touches them), so a clean run means every asm load is covered by a wait on every path; the round-3 fault
(epilogue index arithmetic scheduled into a staging register above the drain wait) is exactly one such report.

  python tools/check_async_loads.py [libflowers_hip.so] [--kernels REGEX]
exit status 1 when anything is reported."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LDS_DMA = frozenset({-1})  # marker "destination" of global_load_lds / buffer_load ... lds (no VGPR is written)
MAX_VISITS = 12  # distinct in-flight states followed per label (loops converge after two or three)
VMEM = re.compile(r"^(global|buffer|flat|scratch)_")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def code_objects(lib, tmp):
    """the gfx950 code objects bundled in a host shared library (or the file itself when it already is one)."""
    with open(lib, "rb") as f:
        head = f.read(20)
    if head[18:20] == b"\xe0\x00":  # e_machine EM_AMDGPU
        return [lib]
    link = os.path.join(tmp, "lib.so")
    os.symlink(os.path.abspath(lib), link)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", link], cwd=tmp, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n)


def kernels(obj):
    """name -> list of (label or None, mnemonic, operand text)"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--symbolize-operands", obj],
                         check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    out, cur, label = {}, None, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if m.group(1).startswith("L") and m.group(1)[1:].isdigit() and cur is not None:
                label = m.group(1)
            else:
                cur, label = [], None
                out[m.group(1)] = cur
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        parts = ins.split(None, 1)
        cur.append((label, parts[0], parts[1] if len(parts) > 1 else ""))
        label = None
    return out


def check(name, ins):
    labels = {lab: i for i, (lab, _, _) in enumerate(ins) if lab}
    reports, seen, visits = [], set(), {}
    work = [(0, ())]  # (instruction index, in-flight tuple of (issue index, frozenset dest regs))
    while work:
        pc, fl = work.pop()
        fl = list(fl)
        while pc < len(ins):
            lab, op, args = ins[pc]
            if lab:
                key = (pc, tuple(d for _, d in fl))
                if key in seen or visits.get(pc, 0) >= MAX_VISITS:
                    break
                seen.add(key)
                visits[pc] = visits.get(pc, 0) + 1
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", args)
                if m:
                    n = int(m.group(1))
                    fl = fl[len(fl) - n:] if n else []
            elif op == "s_endpgm":
                # a wave must not end with an LDS-DMA in flight: the block's LDS may be another block's by the time it lands
                for at, dest in fl:
                    if dest is LDS_DMA:
                        reports.append((pc, at))
                break
            else:
                used = vregs(args)
                if VMEM.match(op) and "_load" in op and "_lds" not in op:
                    used = vregs(args.split(",", 1)[1])  # a later load INTO the same registers retires behind the earlier one: fine
                for at, dest in fl:
                    if dest & used:
                        reports.append((pc, at))
                if VMEM.match(op):
                    dest = LDS_DMA if ("_load" in op and "_lds" in op) or (" lds" in " " + args) else frozenset()
                    if "_load" in op and "_lds" not in op and dest is not LDS_DMA or "atomic" in op and " glc" in args:
                        dest = frozenset(vregs(args.split(",")[0]))
                    fl.append((pc, dest))
                    fl = fl[-64:]
                if op == "s_branch":
                    t = args.split()[0]
                    if t not in labels:
                        break
                    pc = labels[t]
                    continue
                if op.startswith("s_cbranch"):
                    t = args.split()[0]
                    if t in labels:
                        work.append((labels[t], tuple(fl)))
            pc += 1
    uniq = sorted(set(reports))
    return [(pc, at) for pc, at in uniq]


def main():
    args = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--kernels"]
    here = os.path.dirname(os.path.abspath(__file__))
    lib = args[0] if args else os.path.join(here, "..", "oxford_102_flower_gan_vae_latent_diffusion_amd", "libflowers_hip.so")
    pat = re.compile(sys.argv[sys.argv.index("--kernels") + 1]) if "--kernels" in sys.argv else None
    bad = total = 0
    with tempfile.TemporaryDirectory() as tmp:
        for obj in code_objects(lib, tmp):
            for name, ins in kernels(obj).items():
                if pat and not pat.search(name):
                    continue
                total += 1
                for pc, at in check(name, ins):
                    bad += 1
                    try:
                        demangled = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip() or name
                    except OSError:
                        demangled = name
                    print("%s\n    +%d  %s %s\n    touches a register of (or ends under) the load still in flight from +%d  %s %s" %
                          (demangled, pc, ins[pc][1], ins[pc][2], at, ins[at][1], ins[at][2]))
    print("%d kernels checked, %d reports" % (total, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
