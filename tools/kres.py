#!/usr/bin/env python3
"""Print VGPR / spill / scratch / occupancy of every kernel in kernels.hip (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
flags = subprocess.run(["make", "-s", "-C", str(ROOT), "print-hipflags"], capture_output=True, text=True, check=True).stdout.split()
cmd = ["/opt/rocm/bin/hipcc", *flags, f"-I{ROOT}/include", "--cuda-device-only",
       "-c", str(ROOT / "rbrt_amd/csrc/kernels.hip"), "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, r in rows.items():
    if flt in name:
        print(f"{name[:70]:70s} VGPR {r.get('VGPRs')} spillV {r.get('VGPRs Spill')} spillS {r.get('SGPRs Spill')} "
              f"scratch {r.get('ScratchSize [bytes/lane]')} occ {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')}")
