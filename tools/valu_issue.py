#!/usr/bin/env python3
"""Builds profiles/r03_c3_valu_issue.json - the secondary `valu_issue` roofline bench.py attaches to its C3 line.

Claim being made checkable (round-2 verdict, item 1): the C3 frame kernel is bound by VALU ISSUE, not by HBM.
  * dynamic instruction counts per wave and frame: rocprofv3 --pmc SQ_INSTS_VALU and its class counters
    (profiles/r03_pmc.txt; the three classes the counters do not separate - half-wave swaps, selects, max - are
    counted in the disassembly of the loop body, tools/isa_hist.py)
  * issue cost per class: tools/ubench/valu_rate2 on the same box (profiles/r03_ubench_valu.txt), ns per
    wave-instruction per SIMD with the SIMD saturated
  * sum over classes = VALU issue time per wave and frame; x 4 waves per SIMD = per frame slot of a CU
  * cross-check: the timing-only build without LDS exchange and dB stores (-DTDSA_ABLATE=6, profiles/r03_c3_ablation.txt)
"""
import json
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
inp = json.load(open(os.path.join(ROOT, "profiles", "r03_c3_valu_inputs.json")))
pw, ns, abl = inp["per_wave_frame"], inp["ubench_ns_per_wave_instr_per_simd"], inp["ablation_us_per_step"]

# split of the counters' FMA class into v_fma_f32 (VOP3: three VGPR operands / negated operand) and v_fmac + v_fmamk
# (VOP2, literal twiddles), and the classes without a counter of their own: disassembly of
# spectrum_kernel<14,false,1,false>, executed path of the C3 mode (tools/isa_hist.py; the loop body holds 246 v_fma
# of which 46 sit in the tracked-DC / near-silent-frame / other-output branches C3 does not take)
fma = pw["SQ_INSTS_VALU_FMA_F32"]
vop3 = 200.0
swaps, selects, vmax, dot4, dpp = 48.0, 32.0, 16.0, 16.0, 18.0
int_other = max(0.0, pw["SQ_INSTS_VALU_INT32"] - dot4 - dpp)
counted = sum(pw[k] for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_CVT",
                              "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_TRANS_F32"))
rest = pw["SQ_INSTS_VALU"] - counted - swaps - selects - vmax
classes = [
    ("v_add_f32 / v_sub_f32", pw["SQ_INSTS_VALU_ADD_F32"], ns["v_sub_f32 v,v"], "SQ_INSTS_VALU_ADD_F32"),
    ("v_mul_f32", pw["SQ_INSTS_VALU_MUL_F32"], ns["v_mul_f32 v,v"], "SQ_INSTS_VALU_MUL_F32"),
    ("v_fmac_f32 / v_fmamk_f32 (VOP2, literal twiddles)", fma - vop3, ns["v_fmamk_f32 literal"], "SQ_INSTS_VALU_FMA_F32 - the 200 below"),
    ("v_fma_f32 (VOP3)", vop3, 1.30, "disassembly, executed path; cost: profiles/r01_ubench_issue.txt"),
    ("v_cvt_f32_ubyteN", pw["SQ_INSTS_VALU_CVT"], ns["v_cvt_f32_ubyte1"], "SQ_INSTS_VALU_CVT"),
    ("v_dot4_u32_u8 (exact DC sums)", dot4, ns["v_dot4_u32_u8"], "disassembly (part of SQ_INSTS_VALU_INT32)"),
    ("v_add_u32_dpp (wave reduce)", dpp, ns["v_add_u32_dpp quad_perm"], "disassembly (part of SQ_INSTS_VALU_INT32)"),
    ("other integer", int_other, ns["v_add_u32"], "SQ_INSTS_VALU_INT32 - the two above"),
    ("v_permlane32_swap_b32 (half-thread exchange)", swaps, ns["v_permlane32_swap_b32"], "disassembly"),
    ("v_cndmask_b32_e64 (the -i of W_32^(u+8))", selects, ns["v_cndmask_b32_e64 (vcc)"], "disassembly"),
    ("v_max_f32 (hold trace)", vmax, ns["v_max_f32 v,v"], "disassembly"),
    ("v_xor / v_mov / rest", rest, ns["v_mov_b32"], "SQ_INSTS_VALU - everything above"),
    ("v_log_f32", pw["SQ_INSTS_VALU_TRANS_F32"], 0.0, "SQ_INSTS_VALU_TRANS_F32; runs beside the VALU (in-situ: removing the 16 logs changes nothing)"),
]
issue_ns = sum(c * t for _, c, t, _ in classes)
slots_batch = -(-8 * 2440 // 256)                   # frame slots of the slowest CU in an 8-step launch
slot_meas = abl["abl6_batch8"] * 8 / slots_batch    # us per frame slot of the VALU-only build
frames_per_cu = 2440 / 256.0
out = {
    "bound": "valu_issue",
    "kernel": "spectrum_kernel<14,false,1,false> (C3)",
    "insts_valu_per_wave_frame": pw["SQ_INSTS_VALU"],
    "classes": [{"class": n, "per_wave_frame": round(c, 1), "ns_per_wave_instr_per_simd": t, "count_from": s} for n, c, t, s in classes],
    "issue_ns_per_wave_frame": issue_ns,
    "waves_per_simd": 4,
    "issue_us_per_frame_slot": 4 * issue_ns * 1e-3,
    "measured_valu_only_us_per_frame_slot": slot_meas,
    "measured_from": "timing-only build without LDS exchange and dB stores (-DTDSA_ABLATE=6), 8-step launches: "
                     f"{abl['abl6_batch8']} us per step x 8 / {slots_batch} frame slots (profiles/r03_c3_ablation.txt)",
    "frame_slots_per_step": frames_per_cu,
    "floor_us_per_step": frames_per_cu * 4 * issue_ns * 1e-3,
    "unit": "us per 2440-frame step",
    "hbm_frac_if_only_valu_issue_remained": 2440 * 81920 / (frames_per_cu * 4 * issue_ns * 1e-9) / 8e12,
    "sources": ["profiles/r03_pmc.txt", "profiles/r03_ubench_valu.txt", "profiles/r03_c3_ablation.txt", "tools/valu_issue.py"],
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_c3_valu_issue.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "classes"}, indent=1))
for n, c, t, _ in classes:
    print(f"  {n:48s} {c:7.1f} x {t:5.2f} ns = {c * t:7.1f} ns")
