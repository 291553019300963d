"""isa_by_line.py FILE.s KERNEL_SUBSTRING [--top N] -- a compiler listing (hipcc -gline-tables-only --save-temps) broken down by
source line: for the kernel whose symbol contains KERNEL_SUBSTRING, the number of vector / scalar / memory instructions every
(file, line) of the sources compiled to and their SIMD time by profiles/r03/issue_costs.json's 4-wave prices (coarse classes)."""
import collections
import re
import sys

COST = {"plain": 2.0, "fma": 3.6, "half": 4.0, "trans": 7.6}
PLAIN = ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32",
         "v_subrev_u32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_subrev_f32", "v_not_b32", "v_accvgpr")
TRANS = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_permlane", "v_sin_", "v_cos_")


def cls(op):
    if op.endswith("_dpp") or "_dpp" in op:
        return "half"
    if op.startswith(PLAIN) and not op.startswith("v_mov_b64"):
        return "plain"
    if op.startswith("v_fma_f32") or op.startswith("v_fmac_f32"):
        return "fma"
    if op.startswith(TRANS):
        return "trans"
    return "half"


def main():
    path, want = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    files, cur, inside = {}, None, False
    by = collections.defaultdict(lambda: collections.Counter())
    for ln in open(path):
        s = ln.strip()
        m = re.match(r"\.file\s+(\d+)\s+(?:\"([^\"]*)\"\s+)?\"([^\"]*)\"", s)
        if m:
            files[int(m.group(1))] = m.group(3).rsplit("/", 1)[-1]
            continue
        if re.match(r"^[\w.$]+:", s) and not s.startswith(".L") and not s.startswith("BB"):
            inside = want in s
            continue
        if not inside:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = by[cur]
        if op.startswith("v_"):
            c["valu"] += 1
            c["cyc"] += COST[cls(op)]
            if op.startswith(("v_cmp", "v_cndmask")):
                c["cmpsel"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    tot = collections.Counter()
    for c in by.values():
        tot.update(c)
    print("total", dict(tot))
    for k, c in sorted(by.items(), key=lambda kv: -kv[1]["cyc"])[:top]:
        print("%-18s %5d  valu %4d  cyc %6.0f (%4.1f%%)  cmp/sel %3d  salu %3d  lds %3d  vmem %3d" % (
            k[0] if k else "?", k[1] if k else 0, c["valu"], c["cyc"], 100 * c["cyc"] / tot["cyc"], c["cmpsel"], c["salu"], c["lds"], c["vmem"]))


main()
