"""Hash of the sources that decide what the trace kernel does per launch (instructions, traffic, work counters).

profiles/*_pmc_summary.json record it next to the rocprofv3 counters; bench.py reports counter-derived figures
(roofline.traffic, measured_hbm, valu_issue) only from a profile whose hash matches the sources it runs, and says
"stale" otherwise -- a kernel change without a fresh profile can then not carry old counters into a new bench line.
"""
from __future__ import annotations

import hashlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL_SOURCES = ("rbrt_amd/csrc/kernels.hip", "rbrt_amd/csrc/megakernel.inl", "rbrt_amd/csrc/device_types.h",
                  "rbrt_amd/csrc/bvh.cpp", "rbrt_amd/csrc/bvh.h", "rbrt_amd/csrc/bvh_device.hip", "rbrt_amd/csrc/bvh_device.h",
                  "rbrt_amd/csrc/api.cpp", "include/rbrt_hip.h", "include/rbrt_hip_debug.h")


def hipflags() -> str:
    """The compiler flags of the product library, from the one place that has them (the Makefile's HIPFLAGS; the A/B and
    resource tools read them through `make -s print-hipflags`): a flag changes the instruction mix like a source edit."""
    lines, on = [], False
    for line in (ROOT / "Makefile").read_text().splitlines():
        if line.startswith("HIPFLAGS"):
            on = True
        if on:
            lines.append(line.rstrip("\\").strip())
            if not line.rstrip().endswith("\\"):
                break
    return " ".join(" ".join(lines).split())


def kernel_source_sha256() -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(rel.encode() + b"\0")
        h.update((ROOT / rel).read_bytes())
    h.update(b"HIPFLAGS\0" + hipflags().encode())
    return h.hexdigest()[:16]
