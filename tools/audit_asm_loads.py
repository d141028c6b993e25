#!/usr/bin/env python3
"""Audit of the hand-issued loads in k_traverse (per-lane mode): between an asm block that issues
global_load_dwordx4 into VGPRs and the asm wait that retires them, the compiler must not read, copy or
spill those registers (it does not know the loads are in flight).  Prints, per kernel instantiation, the
instructions that touch the node-load destinations before the vmcnt(0) wait (expected: none).
Usage: tools/audit_asm_loads.py   (runs hipcc -S on csrc/kernels.hip)"""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
asm = "/tmp/hrt_kernels.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include",
                       f"-I{ROOT}/nvidia-optix-ray-tracer_amd/csrc", "-S", "--cuda-device-only", "-o", asm,
                       str(ROOT / "nvidia-optix-ray-tracer_amd/csrc/kernels.hip")], stderr=subprocess.DEVNULL)
text = open(asm).read()
bad_total = 0
for m in re.finditer(r"^(_ZN3hrt10k_traverse\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
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
    for what, n_loads, wait in (("node", 5, "s_waitcnt vmcnt(0)"), ("prim", 3, "s_waitcnt vmcnt(5)")):
      loads = [b for b in blocks if b[2].count("global_load_dwordx4 v[") == n_loads]
      waits0 = [b for b in blocks if wait in b[2]]
      for lb in loads:
          regs = set()
          for r in re.finditer(r"global_load_dwordx4 v\[(\d+):(\d+)\]", lb[2]):
              regs.update(range(int(r.group(1)), int(r.group(2)) + 1))
          stop = [w for w in waits0 if w[0] > lb[1]]
          if not stop:
              continue
          bad = []
          for k in range(lb[1] + 1, stop[0][0]):
              l = body[k]
              if l.strip().startswith(";"):
                  continue
              for r in re.finditer(r"\bv(\d+)\b|v\[(\d+):(\d+)\]", l):
                  rs = [int(r.group(1))] if r.group(1) else range(int(r.group(2)), int(r.group(3)) + 1)
                  if any(x in regs for x in rs):
                      bad.append(l.strip())
                      break
          bad_total += len(bad)
          print(f"{name}: {what} loads -> {len(regs)} VGPRs, instructions touching them before the wait: {len(bad)}")
          for b in bad[:5]:
              print("    ", b)
sys.exit(1 if bad_total else 0)
