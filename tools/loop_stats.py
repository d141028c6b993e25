#!/usr/bin/env python3
"""Instruction counts of the traversal loops (the inner loop around the hand-issued node loads) of every path / traverse kernel
instantiation, from the assembly the Makefile's flags produce: VALU split into the dual-pipe ("simple") and single-pipe ("complex")
kinds of profiles/r02_valu_pipes_microbench.txt, SALU, memory.  The SIMD issues about one instruction of ANY kind per 2.4 cycles
(profiles/r02_valu_issue_patterns_microbench.txt), so the total is what a loop iteration costs.
Usage: tools/loop_stats.py [substring of the mangled kernel name ...]"""
import re, subprocess, sys
from collections import Counter
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
from audit_asm_loads import makefile_hipflags
SIMPLE = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32",
          "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_mov_b64", "v_cndmask_b32", "v_not_b32"}


def loops(source, prefix):
    asm = f"/tmp/hrt_loops_{Path(source).stem}.s"
    flags = [f for f in makefile_hipflags() if f != "-fPIC"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", "-o", asm, source], cwd=ROOT, stderr=subprocess.DEVNULL)
    text = open(asm).read()
    for m in re.finditer(r"^(" + prefix + r"\w+):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
        loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l and "offset:64" in l]
        if not loads:
            continue
        idx = loads[0]
        hdrs = [i for i, l in enumerate(body) if "Loop Header: Depth=2" in l and i < idx]
        if not hdrs:
            continue
        hdr = max(hdrs)
        nxt = [i for i, l in enumerate(body) if ("Loop Header: Depth=1" in l or "Loop Header: Depth=2" in l) and i > idx]
        limit = min(nxt) if nxt else len(body)
        back = [(i, labels[l.strip().split()[-1]]) for i, l in enumerate(body)
                if idx < i < limit and "branch" in l and l.strip().split()[-1] in labels and hdr - 60 <= labels[l.strip().split()[-1]] <= hdr]
        if not back:
            continue
        end, start = back[-1]
        cs, ops = Counter(), Counter()
        for l in body[start:end + 1]:
            t = l.strip().split()
            if not t or t[0].startswith(";") or t[0].endswith(":"):
                continue
            op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
            ops[op] += 1
            if op.startswith("v_"):
                cs["valu_simple" if op in SIMPLE else "valu_complex"] += 1
            elif op.startswith("s_"):
                cs["salu"] += 1
            else:
                cs["mem"] += 1
        yield name, cs, ops


if __name__ == "__main__":
    want = sys.argv[1:]
    for src, prefix in (("nvidia-optix-ray-tracer_amd/csrc/kernels.hip", "_ZN3hrt10k_traverse"), ("nvidia-optix-ray-tracer_amd/csrc/fused.hip", "_ZN3hrt7k_fused")):
        for name, cs, ops in loops(src, prefix):
            if want and not any(w in name for w in want):
                continue
            total = sum(cs.values())
            print(f"{name}: {total} instructions in the traversal loop: VALU {cs['valu_simple']} dual-pipe + {cs['valu_complex']} single-pipe, SALU {cs['salu']}, memory {cs['mem']}")
            if "-v" in want or len(want) == 1:
                print("   ", ", ".join(f"{k} {v}" for k, v in ops.most_common(24)))
