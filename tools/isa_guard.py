#!/usr/bin/env python3
"""Resource check of the timed kernel instances, from the compiler's own assembly (closes ADVICE r03 #1).

usage: isa_guard.py [--check] [-DFOO=1 ...]

Compiles cpu_tsdf_amd/csrc/tsdf_integrate.hip to assembly with the build's flags (plus any -D given) and reports, for
the instances the bench times, registers, scratch bytes, occupancy and -- what decides the kernel's speed -- whether a
register spill is RELOADED inside the row loop: a scratch reload there is a vector memory operation whose wait also covers
every voxel store in flight (in-order counter), which cost 15 % in round 5 and 12 % in round 4 each time it crept in.
--check exits non-zero when an instance misses its budget: `__graft_entry__.build()` runs it after compiling (and
tests/test_clean_build.py on the CPU tier), so a toolchain bump cannot silently change spills or occupancy."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# mangled-name prefix -> (what, max VGPRs, max scratch bytes, least occupancy)
BUDGET = {
    "_ZL11k_integrateILi0ELb1ELb1ELb0ELb1ELb1ELb0E": ("k_integrate ALLIN PACKED colour (headline)", 64, 16, 8),
    "_ZL11k_integrateILi0ELb0ELb1ELb0ELb1ELb1ELb0E": ("k_integrate ALLIN PACKED no colour", 64, 16, 8),
    "_ZL13k_integrate_pILi0ELb0EE": ("k_integrate_p (software-pipelined rows, no colour)", 64, 16, 8),
    "_ZL14k_integrate_pcILi0ELb0EE": ("k_integrate_pc (software-pipelined rows, colour: opt-in)", 96, 0, 5),
    "_ZL12k_integrate2ILi0ELb1ELb0E": ("k_integrate2 colour", 96, 0, 5),
    "_ZL12k_integrate2ILi0ELb0ELb0E": ("k_integrate2 no colour", 96, 0, 5),
}


def assemble(flags):
    from cpu_tsdf_amd import build as b
    src = os.path.join(b.CSRC, "tsdf_integrate.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "integrate.s")
        cmd = [b._hipcc()] + b.HIPCC_FLAGS + list(flags) + ["-I" + os.path.join(ROOT, "include"), "-I" + b.CSRC, "-S", "--cuda-device-only", src, "-o", out]
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        # hipcc warns about the unused --hip-link on every -S run: drop that line, show everything else (a real
        # compile error used to vanish in DEVNULL)
        err = "\n".join(ln for ln in r.stderr.splitlines() if "argument unused during compilation" not in ln)
        if err.strip():
            print(err, file=sys.stderr)
        if r.returncode:
            raise SystemExit(f"isa_guard: hipcc -S failed with exit code {r.returncode}")
        return open(out).read().splitlines()


def inspect(lines, prefix):
    start = next((i for i, ln in enumerate(lines) if ln.startswith(prefix) and ln.rstrip().endswith(prefix) is False and ":" in ln), None)
    if start is None:
        return None
    # the kernel's text ends at its .Lfunc_end label (a kernel may hold more than one s_endpgm: early exits); the resource
    # comments (NumVgprs, ScratchSize, Occupancy ...) follow it
    end = next((i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")), None)
    if end is None:
        end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    meta = "\n".join(lines[end:end + 120])

    def num(key):
        m = re.search(r"; " + key + r": (\d+)", meta)
        return int(m.group(1)) if m else -1
    # the row loop: the depth-1 loop that holds the voxel stores
    heads = [i for i, ln in enumerate(body) if "Loop Header: Depth=1" in ln]
    stores = [i for i, ln in enumerate(body) if "buffer_store_dwordx4" in ln or "buffer_store_dword " in ln]
    in_loop_reloads = []
    if heads and stores:
        head = max(h for h in heads if h < stores[0])
        tail = next((h for h in heads if h > stores[-1]), len(body))
        # the loop's own extent where the compiler's block comments give it: the last block marked "in Loop: Header=<this loop>"
        # (a reload BEHIND the loop -- a value parked across it for the epilogue -- is no vector-memory operation of the row path)
        m = re.match(r"\.(LBB\d+_\d+):", body[head])
        if m:
            inside = [i for i in range(head, tail) if ("Header=" + m.group(1)[1:] + " ") in body[i]]
            if inside:
                nxt = next((i for i in range(max(inside) + 1, tail) if re.match(r"\.LBB\d+_\d+:", body[i]) or body[i].startswith("; %bb.")), tail)
                tail = min(tail, nxt)
        for i in range(head, tail):
            if "scratch_load" in body[i]:
                # a reload under a wave-uniform rare branch is harmless; one on the path every row takes is not.  Heuristic: the
                # reloads of the rare in-band blocks sit between an s_cbranch_execz and a ds_write_b8 (the band flag)
                window = "\n".join(body[i:i + 12])
                in_loop_reloads.append((i, "ds_write_b8" in window))
    lanes = sum(1 for ln in body if re.match(r"\s+v_(readlane|writelane)_b32", ln))
    return {"vgprs": num("NumVgprs"), "sgprs": num("TotalNumSgprs"), "scratch": num("ScratchSize"), "occupancy": num("Occupancy"),
            "lds": num("LDSByteSize"), "instructions": sum(1 for ln in body if re.match(r"\s+[a-z]", ln)),
            "hot_reloads": [i for i, rare in in_loop_reloads if not rare], "rare_reloads": [i for i, rare in in_loop_reloads if rare],
            "lane_spill_ops": lanes}


def main():
    check = "--check" in sys.argv
    flags = [a for a in sys.argv[1:] if a != "--check"]
    lines = assemble(flags)
    bad = []
    for prefix, (what, max_v, max_s, occ) in BUDGET.items():
        r = inspect(lines, prefix)
        if r is None:
            print(f"{what}: instance not found")
            bad.append(what)
            continue
        ok = r["vgprs"] <= max_v and r["scratch"] <= max_s and r["occupancy"] >= occ and not r["hot_reloads"]
        print(f"{'ok ' if ok else 'BAD'} {what}: {r['vgprs']} VGPR (<= {max_v}), {r['sgprs']} SGPR, scratch {r['scratch']} B (<= {max_s}), "
              f"occupancy {r['occupancy']} (>= {occ}), LDS {r['lds']} B, {r['instructions']} instructions, spill reloads on the row path "
              f"{len(r['hot_reloads'])}, in rare blocks {len(r['rare_reloads'])}, SGPR lane moves {r['lane_spill_ops']}")
        if not ok:
            bad.append(what)
    if check and bad:
        raise SystemExit("isa_guard: over budget: " + "; ".join(bad))


if __name__ == "__main__":
    main()
