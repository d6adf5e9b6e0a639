#!/usr/bin/env python3
"""Instruction mix of one kernel instance from the compiler's own assembly (hipcc --save-temps): counts by class for
the whole function and for its hot loop (first loop header to the last back edge), plus the resource summary.
usage: isa_mix.py <file.s> <mangled-name-prefix> [<label of the hot loop header>]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "VALU packed fp32 (v_pk_*)"
    if op.startswith(("v_cmp", "v_cmpx")):
        return "VALU compare"
    if op.startswith(("v_cvt", "v_fract", "v_rcp", "v_rsq", "v_sqrt")):
        return "VALU convert / transcendental"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "VALU lane access (SGPR spill traffic)"
    if op.startswith("v_"):
        return "VALU other"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait / barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SMEM"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(prefix) and ln.rstrip().endswith(":") or (ln.startswith(prefix) and ":" in ln))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    loops = [i for i, ln in enumerate(body) if "Loop Header: Depth=1" in ln]
    back = [i for i, ln in enumerate(body) if re.match(r"\s*s_c?branch\w*\s+\.LBB\d+_\d+", ln)]

    def count(seg):
        c = collections.Counter()
        for ln in seg:
            m = re.match(r"\s+([a-z_0-9]+)\s", ln + " ")
            if m and not ln.strip().startswith((";", ".")):
                c[classify(m.group(1))] += 1
        return c
    print(f"kernel {prefix}")
    for name, seg in (("whole function", body), ("outer loop body (hot path + its rare branches)", body[loops[0]:back[-1] + 1] if loops else [])):
        c = count(seg)
        tot = sum(c.values())
        print(f"\n{name}: {tot} instructions")
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print(f"  {v:6d}  {100.0 * v / max(tot, 1):5.1f} %  {k}")
    meta = [ln for ln in lines[end:end + 120] if re.match(r"; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|sgpr_spill_count|vgpr_spill_count|SGPRBlocks|VGPRBlocks)", ln)]
    print("\n" + "\n".join(meta[:9]))
    txt = "\n".join(lines[end:end + 200])
    for key in ("sgpr_spill_count", "vgpr_spill_count"):
        m = re.search(key + r":\s*(\d+)", txt)
        if m:
            print(f"; {key}: {m.group(1)}")


if __name__ == "__main__":
    main()
