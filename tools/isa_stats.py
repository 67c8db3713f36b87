#!/usr/bin/env python3
"""Static instruction statistics of the kernels in one .hip file (no GPU needed).

    python tools/isa_stats.py runia_core_amd/csrc/fused.hip [name-filter]

Compiles the file to gfx950 assembly with the Makefile's flags and prints, per kernel: vector / scalar instruction
counts, v_readlane + v_writelane (scalar registers spilled to vector lanes), MFMA count, VGPRs, SGPRs, scratch bytes.
The hot kernels are sensitive to small source changes (K1 went from 722 to 937 vector instructions when one more
scalar load entered its loop), so this is run before a GPU measurement."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = (__import__("os").environ.get("ISA_EXTRA", "") + " --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form").split()


def main():
    src = Path(sys.argv[1]).resolve()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "k.s"
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", f"-I{ROOT}/include", str(src), "-o", str(out)],
                       check=True, stderr=subprocess.DEVNULL)
        text = out.read_text()
    meta = {}
    kernels = text[text.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in text else ""
    for block in re.split(r"\n  - \.", kernels)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", block)
        if not nm:
            continue
        def g(k, block=block):
            r = re.search(rf"{k}:\s+(\d+)", block)
            return int(r.group(1)) if r else 0
        meta[nm.group(1)] = (g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
    rows = []
    for name, (vg, sg, scratch, lds) in meta.items():
        m = re.search(rf"^{re.escape(name)}:[^\n]*\n(.*?)s_endpgm", text, re.S | re.M)
        if not m:
            continue
        body = m.group(1)
        ins = re.findall(r"^\s+([vs]_[a-z0-9_]+)", body, re.M)
        valu = sum(1 for i in ins if i.startswith("v_") and "mfma" not in i)
        mfma = sum(1 for i in ins if "mfma" in i)
        salu = sum(1 for i in ins if i.startswith("s_"))
        lanes = sum(1 for i in ins if i.startswith(("v_readlane", "v_writelane")))
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = short.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt and flt not in short:
            continue
        rows.append((short, valu, salu, lanes, mfma, vg, sg, scratch, lds))
    print(f"{'kernel':64s} {'valu':>5s} {'salu':>5s} {'lane':>5s} {'mfma':>5s} {'vgpr':>5s} {'sgpr':>5s} {'scr':>5s} {'lds':>6s}")
    for r in rows:
        print(f"{r[0][:64]:64s} " + " ".join(f"{v:5d}" for v in r[1:8]) + f" {r[8]:6d}")


if __name__ == "__main__":
    main()
