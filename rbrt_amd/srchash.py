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
                  "rbrt_amd/csrc/bvh.cpp", "rbrt_amd/csrc/api.cpp")


def kernel_source_sha256() -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update(rel.encode() + b"\0")
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()[:16]
