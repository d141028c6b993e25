#!/usr/bin/env python3
"""Audit of the hand-issued loads of the traversal kernels (k_traverse in kernels.hip, k_fused in fused.hip, k_trace_queue in fused_queue.hip): between an asm
block that issues global_load_dwordx4 into VGPRs and the asm wait that retires them, the compiler must not read, copy or spill
those registers -- it does not know the loads are in flight, and the hardware does not interlock.  Compiles both files to
assembly with the Makefile's own flags (`make -pn`), then prints, per kernel instantiation, the instructions that touch the
destinations of a group of loads before the wait that covers it (expected: none).  The scan follows the text order of the
assembly, which for these loops (issue at the top of the body, waits further down the same body) is the order of execution.
Run by tests/test_host_cpu.py; exit status 1 when anything is found.
Usage: tools/audit_asm_loads.py [-v]"""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent


def makefile_hipflags():
    out = subprocess.run(["make", "-pn", "-C", str(ROOT)], capture_output=True, text=True).stdout
    m = re.search(r"^HIPFLAGS := (.*)$", out, re.M)
    if not m:
        raise SystemExit("HIPFLAGS not found in the Makefile")
    csrc = "nvidia-optix-ray-tracer_amd/csrc"
    return m.group(1).replace("$(ARCH)", "gfx950").replace("$(CSRC)", csrc).split()


def audit(source, kernel_prefix, verbose=False):
    asm = f"/tmp/hrt_audit_{Path(source).stem}.s"
    flags = [f for f in makefile_hipflags() if f != "-fPIC"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", asm, source], cwd=ROOT, stderr=subprocess.DEVNULL)
    text = open(asm).read()
    bad_total = n_groups = 0
    for m in re.finditer(r"^(" + kernel_prefix + r"\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        blocks, i = [], 0
        while i < len(body):
            if "#ASMSTART" in body[i]:
                j = i
                while "#ASMEND" not in body[j]:
                    j += 1
                blocks.append((i, j, "\n".join(body[i:j + 1])))
                i = j
            i += 1
        in_asm = set()
        for b in blocks:
            in_asm.update(range(b[0], b[1] + 1))
        for what, n_loads, wait in (("node", 5, "s_waitcnt vmcnt(0)"), ("prim", 3, "s_waitcnt vmcnt(5)")):
            loads = [b for b in blocks if b[2].count("global_load_dwordx4 v[") == n_loads]
            waits = [b for b in blocks if wait in b[2]]
            for lb in loads:
                regs = set()
                for r in re.finditer(r"global_load_dwordx4 v\[(\d+):(\d+)\]", lb[2]):
                    regs.update(range(int(r.group(1)), int(r.group(2)) + 1))
                stop = [w for w in waits if w[0] > lb[1]]
                if not stop:
                    continue
                n_groups += 1
                bad = []
                for k in range(lb[1] + 1, stop[0][0]):
                    l = body[k]
                    if k in in_asm or l.strip().startswith(";") or "implicit-def" in l:
                        continue
                    for r in re.finditer(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", l):
                        rs = [int(r.group(1))] if r.group(1) else range(int(r.group(2)), int(r.group(3)) + 1)
                        if any(x in regs for x in rs):
                            bad.append(l.strip())
                            break
                bad_total += len(bad)
                if verbose or bad:
                    print(f"{name}: {what} loads -> {len(regs)} VGPRs, instructions touching them before the wait: {len(bad)}")
                for b in bad[:6]:
                    print("    ", b)
    return n_groups, bad_total


if __name__ == "__main__":
    v = "-v" in sys.argv
    total_groups = total_bad = 0
    for src, prefix in (("nvidia-optix-ray-tracer_amd/csrc/kernels.hip", "_ZN3hrt10k_traverse"), ("nvidia-optix-ray-tracer_amd/csrc/fused.hip", "_ZN3hrt7k_fused"), ("nvidia-optix-ray-tracer_amd/csrc/fused_queue.hip", "_ZN3hrt13k_trace_queue")):
        g, b = audit(src, prefix, v)
        print(f"{src}: {g} groups of in-flight loads checked, {b} hazardous instructions")
        total_groups += g; total_bad += b
    sys.exit(1 if total_bad or total_groups == 0 else 0)
