#!/usr/bin/env python3
"""Static instruction counts of the trace megakernel per source region (analysis aid, no GPU needed).

Builds kernels.hip for gfx950 with -DRBRT_MARKERS (asm comments at region boundaries, megakernel.inl) and counts
the VALU / SALU / memory instructions between consecutive markers in the emitted assembly of
trace_megakernel<128, false>. Block layout in the .s follows the source closely but not exactly; read the numbers
as estimates. Multiply by the dynamic counts of rbrt_hip_scene_debug_counters to see where SQ_INSTS_VALU goes.
"""
import collections
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
flags = sys.argv[1:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-DRBRT_MARKERS=1",
       *flags, "-S", "--cuda-device-only", "-o", "/tmp/static_cost.s", str(ROOT / "rbrt_amd/csrc/kernels.hip")]
subprocess.run(cmd, check=True, capture_output=True, cwd="/tmp")
text = Path("/tmp/static_cost.s").read_text().splitlines()
start = next(i for i, l in enumerate(text) if l.startswith("_ZN4rbrt16trace_megakernelILi128ELb0ELb0ELb0EEEvNS_11TraceParamsE:"))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
# Issue cost of one wave64 instruction in SIMD cycles with four resident waves per SIMD (profiles/r03_valu_rate.txt):
# full-rate VALU 2, the other VALU ~3.1, transcendental ~6.1, a scalar instruction ~2 (it takes an issue slot: v_fma +
# s_add pairs cost 4.1 per pair), LDS ~6, vector memory ~4 (issue only; the data path is priced by its own counters).
FULL_RATE = {"v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
             "v_xor_b32", "v_and_b32", "v_or_b32", "v_mov_b32", "v_fmamk_f32", "v_fmaak_f32", "v_not_b32", "v_lshlrev_b32", "v_lshrrev_b32"}
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}


def weight(op):
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    if op.startswith("v_"):
        return 6.1 if base in TRANS else 2.0 if base in FULL_RATE else 3.1
    if op.startswith("s_"):
        return 0.0 if op.startswith(("s_waitcnt", "s_endpgm")) else 2.0
    if op.startswith("ds_"):
        return 6.0
    return 4.0


cur = "prologue"
cyc = collections.Counter()
order, cnt = [], collections.defaultdict(lambda: collections.Counter())
heavy = collections.defaultdict(lambda: collections.Counter())
for l in text[start:end]:
    t = l.strip()
    m = re.match(r";\s*@@(\w+)", t)
    if m:
        cur = m.group(1)
        continue
    if not t or t.startswith((";", ".", "_Z")) or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "v" if op.startswith("v_") else "s" if op.startswith("s_") else "m" if op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")) else "o"
    if cur not in order:
        order.append(cur)
    cnt[cur][kind] += 1
    cyc[cur] += weight(op)
    if kind == "v":
        cnt[cur]["v_full" if weight(op) == 2.0 else "v_trans" if weight(op) == 6.1 else "v_half"] += 1
    if op in ("v_div_scale_f32", "v_sqrt_f32", "v_rcp_f32", "v_div_fixup_f32", "s_cbranch_execz", "s_cbranch_execnz", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_scc0", "s_cbranch_scc1"):
        heavy[cur][op] += 1
tot = collections.Counter()
for r in order:
    c = cnt[r]
    tot.update(c)
    h = heavy[r]
    print(f"{r:14s} valu {c['v']:5d} (full {c['v_full']:4d} half {c['v_half']:4d} trans {c['v_trans']:3d}) salu {c['s']:5d} mem {c['m']:4d}   "
          f"issue cycles {cyc[r]:7.0f}   divs {h['v_div_fixup_f32']:3d} sqrt {h['v_sqrt_f32']:3d} rcp {h['v_rcp_f32']:3d}")
print(f"{'total':14s} valu {tot['v']:5d} (full {tot['v_full']} half {tot['v_half']} trans {tot['v_trans']}) salu {tot['s']:5d} mem {tot['m']:4d}   issue cycles {sum(cyc.values()):.0f}")
