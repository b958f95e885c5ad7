"""Register / LDS / spill table of every kernel of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
python tools/kernel_resources.py dlimgedit_amd/csrc/kernels/gemm.hip [substring filter]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDLIMGEDIT_EXPORTS",
       f"-I{ROOT / 'include'}", f"-I{ROOT / 'dlimgedit_amd' / 'csrc'}", "-x", "hip", "-c", src, "-o", "/tmp/_res.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0]
    if flt not in name:
        continue

    def g(key):
        m = re.search(re.escape(key) + r": (\d+)", b)
        return m.group(1) if m else "?"
    print(f"{name[:100]:100s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} spill {g('VGPRs Spill'):>3s} scratch "
          f"{g('ScratchSize [bytes/lane]'):>4s} occ {g('Occupancy [waves/SIMD]')} lds {g('LDS Size [bytes/block]')}")
