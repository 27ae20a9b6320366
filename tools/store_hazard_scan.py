#!/usr/bin/env python3
"""Static scan of the library's gfx950 code for the store-data hazard of EXPERIMENTS.md R5.13: a vector-memory STORE of 96 / 128 bits whose data
registers are written again within the next few instructions of the same basic block (no s_waitcnt vmcnt(0) in between).  Alone on its SIMD a
wave has sent the store's data by then; with a second wave on the SIMD the input layer's kernel was seen to send the NEW values.  Compiles every
csrc/*.hip to assembly (hipcc -S, device only) and reports, per kernel, the stores with a rewrite of a data register within WINDOW instructions.

    python tools/store_hazard_scan.py [window]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WINDOW = int(sys.argv[1]) if len(sys.argv) > 1 else 6
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]
store_re = re.compile(r"^\s*(global_store_dwordx[34]|buffer_store_dwordx[34]|flat_store_dwordx[34])\s+(.*)$")
reg_re = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in reg_re.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def data_regs(op, rest):
    ops = [t.strip() for t in rest.split(",")]
    # global_store: vaddr, vdata, saddr|off ; buffer_store: vdata, vaddr|off, srsrc, soffset
    return regs(ops[1]) if op.startswith(("global", "flat")) else regs(ops[0])


def dst_regs(line):
    m = re.match(r"^\s*([a-z_0-9]+)\s+(.*)$", line)
    if not m:
        return set()
    op, rest = m.group(1), m.group(2)
    if op.startswith(("s_", "buffer_store", "global_store", "flat_store", "ds_write", "ds_add", ";")):
        return set()
    first = rest.split(",")[0]
    if op.startswith("ds_read2") or op.startswith("ds_read") or op.startswith(("v_", "global_load", "buffer_load", "flat_load")):
        return regs(first)
    return set()


total = 0
for src in sorted(glob.glob(os.path.join(ROOT, "mv3d_tf_amd", "csrc", "*.hip"))):
    asm = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", "-"], capture_output=True, text=True).stdout.splitlines()
    kernel, hits = None, {}
    for i, line in enumerate(asm):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            kernel = m.group(1)
        s = store_re.match(line)
        if not s or kernel is None:
            continue
        data = data_regs(s.group(1), s.group(2))
        n = 0
        for nxt in asm[i + 1:]:
            t = nxt.strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                if t.endswith(":"):
                    break                                  # next basic block
                continue
            if "s_waitcnt" in t and "vmcnt(0)" in t:
                break
            if t.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                break
            n += 1
            if dst_regs(t) & data:
                hits.setdefault(kernel, []).append((i + 1, n, s.group(1), t.split()[0]))
                break
            if n >= WINDOW:
                break
    for k, v in hits.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:90]
        near = min(x[1] for x in v)
        print("%-22s %-92s stores rewritten within %d instr: %3d (nearest: %d)" % (os.path.basename(src), name, WINDOW, len(v), near))
        total += len(v)
print("total:", total)
