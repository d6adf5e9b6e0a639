#!/usr/bin/env python3
"""Instruction budget of the timed k_integrate instance BY PHASE ("the next cut is being chosen blind").

usage: isa_phase_mix.py [-DFOO=1 ...] > profiles/r05_isa_phase_mix.txt

Compiles tsdf_integrate.hip to assembly with line tables (-gline-tables-only: same code, plus .loc directives), takes the
row loop of the headline instance (k_integrate<PCL_SSE, colour, fast projection, !count, PACKED, ALLIN>) and attributes
every instruction to the source line it came from -- the innermost line inside tsdf_integrate.hip of its inlining chain --
and through the `// [phase: ...]` markers in that file to a phase of updateVoxel.  Counts are STATIC (instructions in the
loop body); the phases marked rare sit behind wave-uniform or exec-mask branches that most rows skip, so the second table
weights each phase by how often a wave-row of the 2048^3 headline frame executes it (fractions measured or derived in
DESIGN.md 3.1; they are inputs here, printed with the table)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
INSTANCE = os.environ.get("ISA_INSTANCE", "_ZL11k_integrateILi0ELb1ELb1ELb0ELb1ELb1ELb0E")  # ..ILi0ELb0E.. = without colour
# share of the headline's wave-rows that run a phase (Scene A, 2048^3): everything not listed runs in every row that is not
# left early; 72 % of the wave-rows have an observed voxel (the rest leave after the hinge test)
WEIGHT = {
    "exact fp64 re-projection (rare: uncertified voxels)": 0.28 * 1.15 / 4,  # 28 % of wave-rows enter, ~1.15 of the 4 copies run
    "IEEE fallback (rare)": 0.0005,
    "normalise: raw / neg ladder (rows with an in-band voxel)": 0.10,
    "d update (octree.cpp:152-163)": 0.72 * 0.10,           # a tenth of the observed waves hold an in-band or off-hinge voxel
    "decode count / weight (PACKED)": 0.72,
    "band / implied-distance flags, hinge rest test": 0.72,
    "colour update (octree.cpp:328-337)": 0.72,
    "select / change detection / store": 0.72,
    "transform + project, general instance (not in ALLIN)": 0.0,
}


# SIMD cycles per wave64 instruction, measured on MI355X by tools/ubench/valu_cost.hip (profiles/r06_ubench_valu_cost_call36.txt,
# eight waves per SIMD, 2400 MHz): plain fp32 arithmetic, moves, 32-bit add / sub and two-input logic ~3; packed fp32 ~5.2;
# fp64 5-6; v_rcp_f32 8.6, v_rcp_f64 17; everything else (conversions, compares, shifts, three-input integer, min / max,
# selects, lane access) ~4.5.
def cycles_of(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith("v_pk_"):
        return 5.2
    if base == "v_rcp_f64":
        return 17.0
    if base in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32"):
        return 8.6
    if base.endswith("_f64") or base in ("v_mad_u64_u32", "v_lshl_add_u64"):
        return 5.7
    if base in ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_legacy_f32", "v_mov_b32",
                "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32"):
        return 3.0
    return 4.5


def classify(op):
    if op.startswith("v_pk_"):
        return "VALU"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "VALU"
    if op.startswith("v_"):
        return "VALU"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    from cpu_tsdf_amd import build as b
    src = os.path.join(b.CSRC, "tsdf_integrate.hip")
    text = open(src).read().split("\n")
    markers = [(i + 1, m.group(1)) for i, ln in enumerate(text) for m in [re.search(r"// \[phase: (.+)\]", ln)] if m]

    def phase_of(line):
        name = "(prologue / other)"
        for ln, nm in markers:
            if ln <= line:
                name = nm
        return name
    flags = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([b._hipcc()] + b.HIPCC_FLAGS + flags + ["-gline-tables-only", "-I" + os.path.join(ROOT, "include"), "-I" + b.CSRC, "-S",
                        "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith(INSTANCE) and ":" in ln)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    heads = [i for i, ln in enumerate(body) if "Loop Header: Depth=1" in ln]
    stores = [i for i, ln in enumerate(body) if "buffer_store_dwordx4" in ln or "buffer_store_dword " in ln]
    head = max(h for h in heads if h < stores[0])
    tail = next((h for h in heads if h > stores[-1]), len(body))
    counts = collections.defaultdict(collections.Counter)
    cyc = collections.Counter()  # VALU cycles by phase (cycles_of)
    cur = 0
    for ln in body[head:tail]:
        m = re.match(r"\s+\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            # "; file:line:col @[ file:line:col @[ ... ] ]": innermost first; take the innermost location inside tsdf_integrate.hip
            locs = re.findall(r"(\S+?):(\d+):\d+", ln.split(";", 1)[1] if ";" in ln else "")
            line = next((int(l) for f, l in locs if f.endswith("tsdf_integrate.hip") and int(l) > 0), None)
            if line is None and int(m.group(1)) == 0 and int(m.group(2)) > 0:
                line = int(m.group(2))
            if line:
                cur = line
            continue
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", ln)
        if not m or ln.strip().startswith((";", ".")):
            continue
        counts[phase_of(cur)][classify(m.group(1))] += 1
        if classify(m.group(1)) == "VALU":
            cyc[phase_of(cur)] += cycles_of(m.group(1))
    order = []
    for _, nm in markers:
        if nm not in order and nm in counts:
            order.append(nm)
    for nm in counts:
        if nm not in order:
            order.append(nm)
    print(f"# {os.path.basename(__file__)} {' '.join(flags)}: row loop of k_integrate<PCL_SSE, colour, fast projection, PACKED, ALLIN> (the timed instance)")
    print("# static instruction counts of the loop body by phase; `x share` = share of the headline's wave-rows that execute the phase")
    print(f"{'phase':62s} {'VALU':>5s} {'SALU':>5s} {'VMEM':>5s} {'LDS':>4s} {'wait':>5s}   x share  -> VALU  SALU per wave-row   VALU cycles per wave-row")
    tv = ts = wv = ws = wc = 0.0
    for nm in order:
        c = counts[nm]
        w = WEIGHT.get(nm, 1.0)
        tv += c["VALU"]
        ts += c["SALU"]
        wv += c["VALU"] * w
        ws += c["SALU"] * w
        wc += cyc[nm] * w
        print(f"{nm:62s} {c['VALU']:5d} {c['SALU']:5d} {c['VMEM']:5d} {c['LDS']:4d} {c['wait']:5d}   {w:7.3f}  {c['VALU'] * w:6.1f} {c['SALU'] * w:5.1f}   {cyc[nm] * w:8.1f}")
    print(f"{'total':62s} {int(tv):5d} {int(ts):5d}{'':28s}{wv:6.1f} {ws:5.1f}   {wc:8.1f}")
    print(f"# VALU cycles: the price list of tools/ubench/valu_cost.hip (cycles_of above); {wc:.0f} cycles per wave-row x 32.8 k wave-rows per SIMD "
          f"= {wc * 32768 / 2.4e9 * 1e3:.2f} ms of vector issue per launch at 2.4 GHz")
    print("# measured (rocprofv3 SQ_INSTS_VALU / SQ_INSTS_SALU over the timed launches / 33.55 M wave-rows): see profiles/r05_summary_pmc_SQ2.json")


if __name__ == "__main__":
    main()
