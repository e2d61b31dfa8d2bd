"""Developer tool: static instruction mix of one kernel from a hipcc --save-temps .s file.

  python tools/isa_stats.py <file.s> <kernel-name-substring> [--blocks] [--all]

Prints the register / scratch / LDS figures of the kernel descriptor and, per basic block (label) and in total,
instruction counts by class (VALU f64 / f32 / packed / int, transcendental, DS, VMEM, SALU, waitcnt).  The frame loop of a
kernel is the block (or run of blocks) with the largest count; --blocks lists all blocks with more than 20 instructions.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        if re.match(r"v_(sqrt|rsq|rcp|log|exp|sin|cos)_", op):
            return "trans"
        if op.startswith("v_pk_"):
            return "v_pk"
        if "_f64" in op:
            return "v_f64"
        if "_f32" in op or "_f16" in op:
            return "v_f32"
        if op.startswith(("v_accvgpr", "v_mfma")):
            return "v_acc"
        return "v_int"
    return "other"


def kernels(path):
    cur, body, out = None, [], OrderedDict()
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur, body = m.group(1), []
            out[cur] = body
            continue
        if cur is not None:
            body.append(line.rstrip("\n"))
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                pass
    return out


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    for name, body in kernels(path).items():
        if pat not in name:
            continue
        blocks, cur = OrderedDict(), "entry"
        blocks[cur] = Counter()
        ops_by_block = {cur: Counter()}
        meta = {}
        done = False
        for ln in body:
            s = ln.strip()
            if s.startswith(".Lfunc_end") or s.startswith("s_endpgm") and False:
                done = True
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m and not done:
                cur = m.group(1)
                blocks[cur] = Counter()
                ops_by_block[cur] = Counter()
                continue
            m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs|SGPRSpill|VGPRSpill)\w*: (\d+)", s) or \
                re.match(r"^; (sgpr_spill_count|vgpr_spill_count|NumVgprs|NumAgprs|ScratchSize|Occupancy).*?: (\d+)", s)
            if m:
                meta[m.group(1)] = int(m.group(2))
            if done or not s or s.startswith((";", ".", "//")):
                if s.startswith(".Lfunc_end"):
                    done = True
                continue
            op = s.split()[0]
            if not re.match(r"^[a-z_0-9]+$", op):
                continue
            blocks[cur][classify(op)] += 1
            ops_by_block[cur][op] += 1
        total = Counter()
        for c in blocks.values():
            total.update(c)
        print("==", name)
        print("   ", " ".join("%s=%d" % kv for kv in sorted(meta.items())))
        print("    total:", dict(total), "sum", sum(total.values()))
        big = sorted(blocks.items(), key=lambda kv: -sum(kv[1].values()))
        for lbl, c in (big if show_blocks else big[:3]):
            n = sum(c.values())
            if n < 20:
                break
            valu = sum(v for k, v in c.items() if k.startswith("v_") or k == "trans")
            print("    block %-12s n=%5d valu=%5d  %s" % (lbl, n, valu, dict(c)))
            if "--all" in sys.argv:
                print("        ", ops_by_block[lbl].most_common(40))


if __name__ == "__main__":
    main()
