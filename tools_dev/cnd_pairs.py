#!/usr/bin/env python3
"""cnd_pairs.py <kernel.s> -- count VOP2 v_cndmask_b32 (implicit vcc) instructions that directly follow another one in the
VALU stream (scalar / memory / wait instructions between them do not separate them).  On gfx950 the second select of such a pair
holds its SIMD for ~20 cycles instead of ~4 (tools_dev/ubench/cnd_test.hip); the VOP3 form does not."""
import sys
prev_valu = None
pairs = runs = total = 0
longest = cur = 0
for line in open(sys.argv[1]):
    f = line.split()
    if not f or not f[0].startswith("v_"):
        continue
    m = f[0]
    is_c = m == "v_cndmask_b32_e32" or m == "v_cndmask_b32_dpp" or m == "v_cndmask_b32_sdwa"
    if is_c:
        total += 1
    if is_c and prev_valu:
        pairs += 1
        cur += 1
        longest = max(longest, cur)
    else:
        cur = 0
    prev_valu = is_c
print("%s: %d VOP2 selects, %d of them directly behind another one (longest run %d)" % (sys.argv[1], total, pairs, longest + 1 if pairs else 1))
