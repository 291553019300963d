#!/usr/bin/env python3
"""issue_model.py -- the vector-issue floor of a kernel from measured per-class instruction prices.

    python tools_dev/issue_model.py <costs.json> <object-stem> <kernel-regex> <waves_per_simd> [out.json]

costs.json  output of tools_dev/ubench/issue_cost (SIMD time per wave64 instruction of each class at W waves per SIMD, every
            CU busy, measured on the GPU box).
The kernel is disassembled from wmix_amd/csrc/build/<stem>.o (no GPU needed), every VALU instruction is put into one of the
measured classes, and the histogram is priced at the kernel's occupancy:

    mean_ns_per_valu = sum_class fraction(class) x ns(class, W)

bench.py multiplies that by the DYNAMIC vector-instruction count of a launch (SQ_INSTS_VALU, committed PMC pass) / 1024 SIMDs:
the time the launch would take if its SIMDs did nothing but issue its vector instructions back to back.  The histogram is the
static one of the whole kernel; the kernels it is used for are one straight-line block body executed 2-3 times per launch plus a
short prologue / epilogue, so static and dynamic mixes agree to a few percent (the per-class table is in the output for
inspection).  Round 2 priced every instruction at 4 cycles; VERDICT r02 items 1-2.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = "/opt/rocm/lib/llvm/bin"


def classify(m):
    """mnemonic -> measured class of tools_dev/ubench/issue_cost.hip (None: not a vector-ALU instruction)"""
    if not m.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|sdwa)$", "", m)
    if "_dpp" in base:
        return "v_add_f32_dpp" if re.match(r"v_(add|sub|mul|max|min)", base) else "v_mov_b32_dpp"
    if base.startswith("v_permlane16"):
        return "v_permlane16_swap"
    if base.startswith("v_permlane32"):
        return "v_permlane32_swap"
    if base.startswith(("v_readlane", "v_writelane")):
        return "v_readlane_b32"
    if base.startswith("v_readfirstlane"):
        return "v_readfirstlane_b32"
    if base.startswith(("v_pk_fma", "v_pk_mad")):
        return "v_pk_fma_f32"
    if base.startswith("v_pk_mul"):
        return "v_pk_mul_f32"
    if base.startswith("v_pk_"):
        return "v_pk_add_f32"
    if base.endswith("_f64") or "_f64_" in base:
        if base.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
            return "v_rcp_f64"
        if base.startswith(("v_fma_f64", "v_fmac_f64")):
            return "v_fma_f64"
        if base.startswith(("v_mul_f64", "v_ldexp_f64", "v_div_f")):
            return "v_mul_f64"
        if base.startswith("v_cvt_"):
            return "v_cvt_f64_f32"
        return "v_add_f64"  # add, cmp, min / max, frexp, trunc ...
    for t in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32"):
        if base.startswith(t):
            return t
    if base.startswith(("v_sin_f32", "v_cos_f32", "v_rcp_iflag")):
        return "v_rcp_f32"
    if base.startswith(("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_i32")):
        return "v_mul_lo_u32"
    if base.startswith(("v_mad_u32_u24", "v_mad_i32_i24")):
        return "v_mad_u32_u24"
    if base.startswith(("v_mul_u32_u24", "v_mul_i32_i24")):
        return "v_mul_u32_u24"
    if base.startswith("v_div_scale"):
        return "v_div_scale_f32"
    if base.startswith("v_div_fmas"):
        return "v_div_fmas_f32"
    if base.startswith("v_div_fixup"):
        return "v_div_fixup_f32"
    if base.startswith("v_ldexp_f32"):
        return "v_ldexp_f32"
    if base.startswith(("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")):
        return "v_lshlrev_b64"
    if base.startswith("v_cndmask"):
        return "v_cndmask_b32"
    if base.startswith("v_cmp"):
        return "v_cmp_f32"
    if base.startswith("v_cvt_"):
        return "v_cvt_f32_i32"
    if base.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_mac_f32", "v_fmaak", "v_fmamk")):
        return "v_fma_f32"
    if base.startswith("v_mul_f32"):
        return "v_mul_f32"
    if base.startswith(("v_mov_b32", "v_accvgpr", "v_mov_b64")):
        return "v_mov_b32"
    if base.startswith(("v_max", "v_min", "v_med3")):
        return "v_max_f32"
    if base.startswith(("v_bfe", "v_bfi", "v_alignbit", "v_perm_b32")):
        return "v_bfe_i32"
    if base.startswith(("v_lshl_add", "v_add_lshl", "v_lshl_or", "v_and_or", "v_or3", "v_add3", "v_xad")):
        return "v_lshl_add_u32"
    if base.startswith(("v_and", "v_or", "v_xor", "v_not", "v_lshl", "v_lshr", "v_ashr")):
        return "v_and_b32"
    if base.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_add_i32", "v_sub_i32")):
        return "v_add_u32"
    return "v_add_f32"  # v_add / v_sub / v_subrev f32 and whatever plain single-rate op is left


def disassemble(stem, pattern):
    obj = os.path.join(os.environ.get("WMX_TOOL_OBJDIR", os.path.join(ROOT, "wmix_amd", "csrc", "build")), stem + ".o")
    with tempfile.TemporaryDirectory() as t:
        subprocess.check_call([B + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + t + "/fat.bin", obj])
        subprocess.check_call([B + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + t + "/fat.bin",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + t + "/dev.co"])
        txt = subprocess.check_output([B + "/llvm-objdump", "-d", "--demangle", t + "/dev.co"], text=True)
    out, on, name = [], False, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            on = re.search(pattern, m.group(1)) is not None
            if on:
                name = m.group(1)
            continue
        if on:
            f = line.split()
            if f and re.match(r"^[a-z]", f[0]):
                out.append(f[0])
    return name, out


def model(costs, stem, pattern, waves):
    name, ins = disassemble(stem, pattern)
    ws = costs["waves_per_simd"]
    # a column is usable when the micro-benchmark really had that many waves per SIMD running together (its own check:
    # resident_waves_per_simd); above four its occupancy pinning does not always hold.  Prices fall with occupancy, so the
    # nearest usable column at or below the kernel's occupancy never flatters the kernel.
    res = costs["classes"]["v_add_f32"].get("resident_waves_per_simd")
    ok = [i for i in range(len(ws)) if ws[i] <= waves and (res is None or res[i] >= 0.95 * ws[i])]
    col = max(ok, key=lambda i: ws[i]) if ok else 0
    hist = collections.Counter()
    other = collections.Counter()
    prev_vop2_select = False
    for m in ins:
        c = classify(m)
        if c is not None:
            # a VOP2 select directly behind another one in the vector stream has its own (much higher) measured price
            vop2_select = m in ("v_cndmask_b32_e32", "v_cndmask_b32_dpp", "v_cndmask_b32_sdwa")
            if vop2_select and prev_vop2_select:
                c = "v_cndmask_b32_vop2_behind_vop2"
            prev_vop2_select = vop2_select
        if c is None:
            other["salu" if m.startswith("s_") else ("lds" if m.startswith("ds_") else ("vmem" if m.startswith(("global", "buffer", "flat", "scratch")) else "other"))] += 1
        else:
            hist[c] += 1
    n = sum(hist.values())
    classes = {}
    ns = cyc = 0.0
    for c, k in hist.most_common():
        e = costs["classes"][c]
        classes[c] = {"static_count": k, "fraction": round(k / n, 4), "ns": e["ns"][col], "cycles": e["cycles"][col]}
        ns += k / n * e["ns"][col]
        cyc += k / n * e["cycles"][col]
    # the scalar side: ALU instructions at the price of a pure scalar stream (one scalar ALU serves the CU's four SIMDs), s_nop /
    # s_waitcnt at the price of s_nop; beside vector work they cost less (the mix classes of the table), so this is a ceiling
    salu = [m for m in ins if m.startswith("s_")]
    n_idle = sum(1 for m in salu if m in ("s_nop", "s_waitcnt"))
    n_alu = len(salu) - n_idle
    scalar = {"static_alu_branch_smem": n_alu, "static_nop_waitcnt": n_idle, "ns_alu": costs["classes"]["s_add_u32"]["ns"][col],
              "ns_nop": costs["classes"]["s_nop"]["ns"][col]}
    plain = costs["classes"]["v_add_f32"]
    valu = {c: e for c, e in costs["classes"].items() if c.startswith("v_")}
    cheapest = min(valu, key=lambda c: valu[c]["ns"][col])
    return {"kernel": name, "object": stem + ".o", "waves_per_simd": waves, "priced_at_waves_per_simd": ws[col], "static_valu_instructions": n,
            "static_other_instructions": dict(other), "mean_ns_per_valu": round(ns, 4), "mean_cycles_per_valu": round(cyc, 3),
            "plain_v_add_f32": {"ns": plain["ns"][col], "cycles": plain["cycles"][col]},
            "cheapest_valu": {"class": cheapest, "ns": valu[cheapest]["ns"][col]},
            "scalar": scalar, "classes": classes, "costs_from": costs.get("device", "?"), "costs_method": costs.get("method", "wall clock, v1")}


if __name__ == "__main__":
    costs = json.load(open(sys.argv[1]))
    r = model(costs, sys.argv[2], sys.argv[3], int(sys.argv[4]))
    s = json.dumps(r, indent=1)
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(s + "\n")
    print(s)
