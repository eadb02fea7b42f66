"""Cuts the two traversal steps out of the compiled closest-hit kernel and counts their instructions.

    python profiles/isa_loops.py profiles/r03_k_trace [more compiler flags, e.g. -DMIPT_PK_FMA]

compiles csrc/pt_kernels.hip to assembly with -DMIPT_ISA_MARKERS (the steps are then bracketed by comments) and
writes <prefix>_node_step.s and <prefix>_tri_step.s: the inner loop of k_trace<0, false, false> that holds the marker,
every block of it (also the ones a wave only enters when a lane's stack has left its LDS part), under a header with
the instruction count per category of one trip along the straight path.  "full rate" vector instructions
are the ones that profiles/microbench/valu_throughput.hip measures at about 1 ns of SIMD time per wave64
instruction (v_fma/mul/add/sub/mov/and/or/xor/not, 32-bit integer add and subtract); every other vector
instruction takes about 1.8 ns."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_RATE = ("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
             "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_fmac_f32", "v_not_b32")


def category(op):
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "scalar"
    if op.startswith("ds_"):
        return "LDS (incl. lane exchange)"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vector memory"
    if op.startswith("v_"):
        return "vector, full rate" if op.startswith(FULL_RATE) else "vector, half rate"
    return "other"


def split_blocks(kernel):
    """[(label, first line index, [instructions])] of the kernel's basic blocks, in layout order."""
    blocks = []
    for i, l in enumerate(kernel):
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
        if m:
            blocks.append((m.group(1) or "bb." + m.group(2), i, []))
        elif blocks:
            t = l.strip()
            if t and not t.startswith((";", ".")):
                blocks[-1][2].append(t)
    return blocks


def straight_path(blocks, header):
    """Block indices of one trip round the loop that starts at `header`: unconditional branches are followed,
    conditional ones fall through (the compiler lays the expected path out as the fall-through path; the
    branches lead to the blocks for lanes whose stack has left its LDS part, to empty-exec skips and out of the loop)."""
    index = {b[0]: i for i, b in enumerate(blocks)}
    path, i = [], index[header]
    while True:
        path.append(i)
        last = blocks[i][2][-1] if blocks[i][2] else ""
        i = index[last.split()[1]] if last.startswith("s_branch") else i + 1
        if blocks[i][0] == header or len(path) > 200:
            return path


def main(prefix, extra_flags=()):
    src = os.path.join(ROOT, "cuda-path-tracer_amd", "csrc", "pt_kernels.hip")
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                          "-fno-slp-vectorize", "-Wno-unused-function", "-DMIPT_ISA_MARKERS", *extra_flags, "--cuda-device-only", "-S", "-o", "-", src],
                         capture_output=True, text=True, check=True).stdout.split("\n")
    start = next(i for i, l in enumerate(asm) if l.startswith("_Z7k_traceILi0ELb0ELb0E"))
    end = next(i for i in range(start, len(asm)) if "s_endpgm" in asm[i])
    kernel = asm[start:end + 1]
    blocks = split_blocks(kernel)
    for name, tag in (("node_step", "node step"), ("tri_step", "triangle step")):
        mark = next(i for i, l in enumerate(kernel) if f"MIPT {tag} begin" in l)
        owner = max(b for b in range(len(blocks)) if blocks[b][1] <= mark)
        m = re.search(r"Header=(BB\d+_\d+) Depth=2", kernel[blocks[owner][1]])
        header = ".L" + m.group(1) if m else blocks[owner][0]
        member = [b for b in range(len(blocks)) if blocks[b][0] == header or f"Header={header[2:]} " in kernel[blocks[b][1]]]
        path = straight_path(blocks, header)
        counts = {}
        for b in path:
            for ins in blocks[b][2]:
                c = category(ins.split()[0])
                counts[c] = counts.get(c, 0) + 1
        first, last = blocks[member[0]][1], (blocks[member[-1] + 1][1] if member[-1] + 1 < len(blocks) else len(kernel))
        with open(f"{prefix}_{name}.s", "w") as f:
            f.write(f"; {tag} loop of k_trace<0, false, false> (closest hit), gfx950, compiled from csrc/pt_kernels.hip by profiles/isa_loops.py\n")
            f.write("; one trip of the straight path (conditional branches fall through: no lane with its stack outside the LDS part,\n")
            f.write("; no empty exec mask, loop not left) runs the blocks " + " ".join(blocks[b][0] for b in path) + "\n")
            f.write("; instructions on that trip, per category:\n")
            for c in sorted(counts):
                f.write(f";   {c:28s} {counts[c]:4d}\n")
            f.write(f";   {'all':28s} {sum(counts.values()):4d}\n;\n")
            f.write("\n".join(l for l in kernel[first:last] if "ASMSTART" not in l and "ASMEND" not in l) + "\n")
        print(name, counts, sum(counts.values()))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
