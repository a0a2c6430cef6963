#!/usr/bin/env python3
"""Static instruction mix of the wave stage kernel (csrc/mlp_wave.hip), per region of its work-item loop.

With one wave per SIMD a vector instruction costs the f32 MFMA stream 5 - 6 cycles whichever way it is scheduled
(tools/micro/mfma_two_waves.hip), so the count of vector instructions per item IS the kernel's non-MFMA time; this
script is how changes are judged before they go to the GPU (no GPU needed: hipcc -S).

    python tools/isa_vector_count.py [extra hipcc flags] [--deep] [--asm file.s]
    python tools/isa_vector_count.py --phases     (profiler build: counts between its clock reads = per phase of an item)

Regions: item loop head .. MotionNet layer loop | its body | MotionNet tail, SpaceNet encoding, stage1.0 | SpaceNet layer
loop body (x 6 per item, includes the skip segment's copy) | sigma, rgb_net.1, colour head, store.  `vec` = VALU +
v_accvgpr_read/write; also prints scratch size (must be 0) and register counts.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "st-nerf_amd", "csrc", "mlp_wave.hip")


def classify(line):
    line = line.strip()
    if not line or line[0] in ";.":
        return None
    op = line.split()[0]
    if op.endswith(":"):
        return None
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_accvgpr"):
        return "acc_rw"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "DS"
    if op.startswith(("buffer_", "global_", "scratch_")):
        return "VMEM"
    return None


def main():
    args = sys.argv[1:]
    deep = "--deep" in args
    phases = "--phases" in args
    if phases:
        args = [a for a in args if a != "--phases"] + ["-DSTNERF_WAVE_PROF"]
    asm = None
    if "--asm" in args:
        asm = args[args.index("--asm") + 1]
        args = [a for a in args if a not in ("--asm", asm)]
    flags = [a for a in args if a != "--deep"]
    if asm is None:
        asm = os.path.join(tempfile.mkdtemp(prefix="stnerf_isa_"), "mlp_wave.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", *flags,
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(SRC), "-S", "--cuda-device-only", "-o", asm, SRC]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    tag = "ILb1E" if deep else "ILb0E"
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN6stnerf21mlp_wave_stage_kernel" + tag) and ":" in l.split(";")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    for l in lines[end:end + 400]:
        if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy)", l):
            print(l.strip("; "))
    item = next(i for i, l in enumerate(body) if "This Loop Header: Depth=1" in l)
    if phases:
        # straight-line listing order of the item loop: a phase = the code between two clock reads (loop bodies counted once)
        marks = [i for i in range(item, len(body)) if re.match(r"\s*s_memtime", body[i])]
        prev = item
        for n, m in enumerate(marks + [len(body)]):
            c = collections.Counter(k for k in map(classify, body[prev:m]) if k)
            print(f"segment {n:2d} (lines {prev:5d}..{m:5d})  vec {c['VALU'] + c['acc_rw']:5d}   " +
                  "  ".join(f"{k} {c[k]}" for k in ("VALU", "acc_rw", "MFMA", "DS", "VMEM", "SALU", "wait", "nop")))
            prev = m
        return
    inner = [i for i, l in enumerate(body) if "Parent Loop" in l]
    if len(inner) < 2:
        raise SystemExit("expected two inner loops (MotionNet layers, SpaceNet layers)")

    def loop_end(head):
        label = body[head].split(":")[0]
        return next(i for i in range(head + 1, len(body))
                    if re.match(r"\s*s_c?branch\S*\s+" + re.escape(label) + r"\s*$", body[i]))

    m0, s0 = inner[0], inner[-1]
    me, se = loop_end(m0), loop_end(s0)
    total = collections.Counter()
    for label, a, b, mult in (("item head .. MotionNet loop", item, m0, 1), ("MotionNet layer loop body (x 4)", m0, me + 1, 0),
                              ("MotionNet tail + encoding + stage1.0", me + 1, s0, 1), ("SpaceNet layer loop body (x 6)", s0, se + 1, 6),
                              ("sigma + rgb_net.1 + head + store", se + 1, len(body), 1)):
        c = collections.Counter(k for k in map(classify, body[a:b]) if k)
        vec = c["VALU"] + c["acc_rw"]
        print(f"{label:40s} vec {vec:5d}   " + "  ".join(f"{k} {c[k]}" for k in ("VALU", "acc_rw", "MFMA", "DS", "VMEM", "SALU", "wait", "nop")))
        total["vec"] += vec * mult
        total["mfma"] += c["MFMA"] * mult
    print(f"static totals on the background path (MotionNet loop excluded, its head / tail code included): {total['vec']} vector, "
          f"{total['mfma']} MFMA listed (the skip segment's 256 run once, not 6 x)")


if __name__ == "__main__":
    main()
