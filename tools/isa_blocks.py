#!/usr/bin/env python3
"""Basic blocks of one kernel in a gfx950 assembly dump, biggest first: VALU / SALU / LDS / VMEM counts.
    hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -S --cuda-device-only -o x.s cutseq_amd/csrc/cutseq_hip.hip
    python3 tools/isa_blocks.py x.s _ZN5csdev11trim_kernelILb1ELb0ELi0EEEvNS_5KArgsE [min_instructions]
"""
import re
import sys

path, kernel = sys.argv[1], sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip() == "s_endpgm")
blocks, cur = [], ["entry", start, []]
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), i, []]
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur[2].append(t.split()[0])
blocks.append(cur)
tot = {"v": 0, "s": 0, "ds": 0, "mem": 0}
for name, line, ops in blocks:
    for o in ops:
        key = "ds" if o.startswith("ds_") else "v" if o.startswith("v_") else "s" if o.startswith("s_") else "mem"
        tot[key] += 1
print("kernel lines %d..%d, %d blocks, static counts %s" % (start, end, len(blocks), tot))
for name, line, ops in sorted(blocks, key=lambda b: -len(b[2])):
    if len(ops) < min_n:
        break
    v = sum(o.startswith("v_") for o in ops)
    s = sum(o.startswith("s_") for o in ops)
    ds = sum(o.startswith("ds_") for o in ops)
    mem = len(ops) - v - s - ds
    wait = sum(o == "s_waitcnt" for o in ops)
    nop = sum(o in ("s_nop",) for o in ops)
    bit3 = sum(o.startswith("v_bitop3") for o in ops)
    print("%-12s line %6d  total %4d  valu %4d (bitop3 %3d)  salu %3d (waitcnt %2d, nop %2d)  lds %3d  vmem %2d" %
          (name, line + 1, len(ops), v, bit3, s, wait, nop, ds, mem))
