#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every kernel in libprt_hip.so, straight from the compiler
(hipcc -Rpass-analysis=kernel-resource-usage on prt_api.hip with the Makefile's flags + any -D given).

  python tools/kernel_resources.py [-DPRT_X=1 ...] [--all] [--md]

Used by tools/summarize_profile.py so that occupancy claims can be read from profiles/ (VERDICT round 2, item 2)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtracer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-Wno-unused-function"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def resources(defs=()):
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *defs, "-c", "prt_api.hip", "-o", "/dev/null", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in err.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass", ln)
        if not m:
            continue
        key, _, val = m.group(1).partition(":")
        key, val = key.strip(), val.strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        d = re.sub(r"^void ", "", d)
        r["short"] = re.sub(r"\(.*", "", d).replace("prt::", "")
    return rows


def table(rows, every=False):
    keep = rows if every else [r for r in rows if re.match(r"k_(extend|shadow|shade|ao_shade|resolve|generate)", r["short"])]
    out = ["| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/block | waves per SIMD (registers) |", "|---|---:|---:|---:|---:|---:|---:|"]
    for r in keep:
        out.append(f"| `{r['short']}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('ScratchSize [bytes/lane]')} | "
                   f"{r.get('LDS Size [bytes/block]')} | {r.get('Occupancy [waves/SIMD]')} |")
    return "\n".join(out)


if __name__ == "__main__":
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    print(table(resources(defs), "--all" in sys.argv))
