"""Bank model behind DESIGN_HISTORY.md section 5d (a): LDS cycles of the access families of aec_near_kernel's work rows that conflict.

A 16-lane group (lanes with the same lane % 4: g = lane & 3, gl = lane >> 2) owns one transform in row g (or 4 + g) of the wave's
eight work rows; a row holds 64 complex points (two floats each).  Per 64-sample block the kernel issues, on those rows,
  * 24 ds_read_b64 pairs of the inverse transforms' gathers (point rev4(gl) + {0, 32, 16, 48} and its mirror 64 - p): banks
    = (word address) mod 64, conflicts inside each 32-lane half;
  * 16 ds_write_b64 of the transforms' results (point gl + 16 m): banks = word mod 32, conflicts inside each 16-lane group.
(MI355X_MICROARCH.md "LDS": each extra distinct address on a busy bank within a lane group costs one LDS cycle.)  Everything else
the kernel does on these rows is contiguous by lane and conflict-free.  Prints the cycles of one block's 40 such instructions for
row strides 128 ... 150 words, and searches the per-row XOR swizzle of the 64 slots (slot ^ s[row & 3]) that minimises them.

    python tools_dev/lds_bank_model.py          # ideal: 112 cycles; stride 132: 320; 138: 224; best swizzle at stride 128: 124
"""
import itertools


def rev4(x):
    return int("{:04b}".format(x)[::-1], 2)


def cyc_read_b64(addr):
    tot = 0
    for half in range(2):
        banks = {}
        for lane in range(32 * half, 32 * half + 32):
            a = addr(lane)
            for d in range(2):
                banks.setdefault((a + d) % 64, set()).add(a + d)
        tot += max(len(v) for v in banks.values())
    return tot


def cyc_write_b64(addr):
    tot = 0
    for g4 in range(4):
        banks = {}
        for lane in range(16 * g4, 16 * g4 + 16):
            a = addr(lane)
            for d in range(2):
                banks.setdefault((a + d) % 32, set()).add(a + d)
        tot += max(len(v) for v in banks.values())
    return tot


def block_cycles(stride, s=(0, 0, 0, 0)):
    row = lambda g: g * stride  # noqa: E731
    t = 0
    for off in (0, 32, 16, 48):
        own = lambda lane: row(lane & 3) + 2 * ((rev4(lane >> 2) + off) ^ s[lane & 3])  # noqa: E731
        mir = lambda lane: row(lane & 3) + 2 * (((64 - (rev4(lane >> 2) + off)) & 63) ^ s[lane & 3])  # noqa: E731
        t += 3 * (cyc_read_b64(own) + cyc_read_b64(mir))  # two rows of the packed pairs + one of the single transforms
    for m in range(4):
        st = lambda lane: row(lane & 3) + 2 * (((lane >> 2) + 16 * m) ^ s[lane & 3])  # noqa: E731
        t += 4 * cyc_write_b64(st)
    return t


if __name__ == "__main__":
    print("ideal", 3 * 8 * 2 + 16 * 4)
    print("row stride (words) -> cycles:", {f: block_cycles(f) for f in range(128, 152, 2)})
    best = None
    for o in itertools.product(range(64), repeat=3):
        t = block_cycles(128, (0,) + o)
        if best is None or t < best[0]:
            best = (t, (0,) + o)
    print("best XOR swizzle at stride 128: s = %s -> %d cycles" % (best[1], best[0]))
