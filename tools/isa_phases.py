#!/usr/bin/env python
"""Static look at the memory-latency structure of the gfx950 code of a HIP source: per kernel, the number of
"phases" = s_waitcnt vmcnt(..) instructions that follow at least one global/buffer load (or returning atomic) issued
since the previous such wait, in program order, plus registers and LDS.  A kernel on the frame's latency chain should
need few phases; a loop of `load; wait; use` shows up as many.  (Branches are ignored: it is an upper bound per path.)

    python tools/isa_phases.py surfelmeshing_amd/csrc/smx_recon.hip [kernel-substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(src):
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "surfelmeshing_amd", "csrc"),
           "--cuda-device-only", "-S", "-o", out, src] + os.environ.get("SMX_EXTRA_FLAGS", "").split()
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def main():
    src = sys.argv[1]
    want = [a for a in sys.argv[2:] if a != "--seq"]
    seq = "--seq" in sys.argv   # also print the order of loads (L), returning atomics (A), waits (W<n>) and barriers (B)
    asm = open(device_asm(src)).read().split("\n")
    kernels = {}
    cur = None
    for line in asm:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if line.startswith("\t.section") or line.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            kernels[cur].append(line.strip())
    # (one YAML block per kernel under amdhsa.kernels, keys in alphabetical order: collect a block, then file it by .name)
    meta = {}
    block = {}
    for line in asm:
        if re.match(r"\s+- \.", line) and ".args" in line or re.match(r"\s+- \.agpr_count", line):
            if "name" in block:
                meta[block["name"]] = block
            block = {}
        m = re.search(r"^\s+\.name:\s+(\S+)", line)
        if m and "name" not in block:
            block["name"] = m.group(1)
        for key in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size"):
            m2 = re.search(r"\." + key + r":\s+(\d+)", line)
            if m2:
                block[key] = int(m2.group(1))
    if "name" in block:
        meta[block["name"]] = block
    print("%-44s %6s %6s %6s %7s %7s %6s %6s %7s" % ("kernel", "loads", "phases", "atomic", "stores", "barrier", "vgpr", "lds", "scratch"))
    for k, body in kernels.items():
        short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(anonymous namespace\)::", "", short).split("(")[0]
        if want and not any(w in short for w in want):
            continue
        loads = phases = atom = stores = barriers = 0
        pending = False
        trace = []
        for ins in body:
            op = ins.split(" ")[0].split("\t")[0]
            if seq:
                m3 = re.search(r"vmcnt\((\d+)\)", ins)
                if op.startswith(("global_load", "buffer_load", "flat_load")):
                    trace.append("L" + op.split("_")[-1].replace("dword", "d").replace("ubyte", "b").replace("ushort", "h"))
                elif op.startswith(("global_atomic", "flat_atomic")):
                    trace.append("A" if ("sc0" in ins or " glc" in ins) else "a")
                elif op.startswith(("global_store", "flat_store")):
                    trace.append("s")
                elif op == "s_waitcnt" and m3:
                    trace.append("W%s" % m3.group(1))
                elif op == "s_barrier":
                    trace.append("B")
                elif op == "s_endpgm":
                    trace.append("END")
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                loads += 1
                pending = True
            elif op.startswith(("global_atomic", "flat_atomic", "buffer_atomic")):
                atom += 1
                if "sc0" in ins or " glc" in ins:   # returning
                    pending = True
            elif op.startswith(("global_store", "buffer_store", "flat_store")):
                stores += 1
            elif op == "s_barrier":
                barriers += 1
            elif op == "s_waitcnt" and "vmcnt" in ins and pending:
                phases += 1
                pending = False
        m = meta.get(k, {})
        print("%-44s %6d %6d %6d %7d %7d %6s %6s %7s" % (short[:44], loads, phases, atom, stores, barriers,
                                                          m.get("vgpr_count", "?"), m.get("group_segment_fixed_size", "?"),
                                                          m.get("private_segment_fixed_size", "?")))
        if seq:
            out, prev, n = [], None, 0
            for t in trace + [None]:
                if t == prev:
                    n += 1
                    continue
                if prev is not None:
                    out.append(prev if n == 1 else "%s*%d" % (prev, n))
                prev, n = t, 1
            print("   " + " ".join(out))


if __name__ == "__main__":
    main()
