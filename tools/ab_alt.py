#!/usr/bin/env python3
"""A/B of bench.py configurations by ALTERNATION (every A/B as >= 5 alternations with the spread; box-to-box
and placement variance is larger than the 1-2 % steps being chased, and two runs say nothing).

usage: ab_alt.py [--rounds 5] [--out FILE] [--bench "extra bench args"] name=ENV1=v,ENV2=v[,lib=VARIANT] ...
  each configuration is a set of environment variables; lib=NAME points TSDF_HIP_LIB_PATH at
  cpu_tsdf_amd/lib/variants/NAME/libtsdf_hip.so (tools/build_variant.py).  Round i runs every configuration once, in
  order; the table holds kernel_ms (HIP events around each launch, bench.py's roofline.kernel_ms) per run, then
  mean / min / max / spread per configuration and the plane placement class of each process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    rounds, out, extra = 5, None, ""
    cfgs = []
    i = 0
    while i < len(args):
        if args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        elif args[i] == "--out":
            out = args[i + 1]; i += 2
        elif args[i] == "--bench":
            extra = args[i + 1]; i += 2
        else:
            name, _, spec = args[i].partition("=")
            env = {}
            for kv in filter(None, spec.split(",")):
                k, _, v = kv.partition("=")
                if k == "lib":
                    env["TSDF_HIP_LIB_PATH"] = os.path.join(ROOT, "cpu_tsdf_amd", "lib", "variants", v, "libtsdf_hip.so")
                else:
                    env[k] = v
            cfgs.append((name, env)); i += 1
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--extras", "0", "--cpu-baseline", "0",
            "--host-path", "0"] + extra.split()
    res = {n: [] for n, _ in cfgs}
    place = {n: [] for n, _ in cfgs}
    for r in range(rounds):
        for name, env in cfgs:
            try:
                p = subprocess.run(base, env=dict(os.environ, **env), capture_output=True, text=True, timeout=240)
                d = json.loads(p.stdout.strip().splitlines()[-1])
                res[name].append(round(d["roofline"]["kernel_ms"], 3))
                pm = d["config"]["plane_placement"].get("probe_sweep_ms") or [0]
                place[name].append(round(min(pm) if isinstance(pm, list) else pm, 1))
            except Exception as e:  # noqa: BLE001
                res[name].append(None)
                place[name].append(None)
                print(f"# {name} round {r}: failed: {e}", flush=True)
            print(f"round {r} {name}: {res[name][-1]} ms (placement probe {place[name][-1]})", flush=True)
    lines = [f"# bench.py {' '.join(base[2:])}", f"# {rounds} alternations; kernel_ms per run, then mean / min / max / (max - min)"]
    for name, env in cfgs:
        v = [x for x in res[name] if x is not None]
        if v:
            lines.append(f"{name:>16}: {res[name]}  mean {sum(v) / len(v):.3f}  min {min(v):.3f}  max {max(v):.3f}  spread {max(v) - min(v):.3f}"
                         f"   placement {place[name]}   env {env}")
        else:
            lines.append(f"{name:>16}: all runs failed   env {env}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
