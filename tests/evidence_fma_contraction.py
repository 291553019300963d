"""Evidence for DESIGN_HISTORY.md section 5d (round-3 VERDICT item 1c): what fused multiply-adds do to the float AEC.

Not a test (pytest does not collect it).  north_star allows +-1 LSB for the float path and SURVEY section 0 item 8 measured
`-mfma -ffp-contract=fast` at <= 1 LSB on ONE stream over 3 000 frames.  On the GPU the near kernel built with contraction is
5.3 % faster -- and on many streams over long runs it is up to 941 LSB off (gpurun_out/exp1, profiles/r04/aec_contraction_gpu.txt).
This script shows the same thing on the CPU and names the decision that flips: the restatement is built a second time with
`-mfma -ffp-contract=fast` (oracle/Makefile `fma`), the AEC of the 3 000-frame parity gate's streams is run packet by packet
in both builds on IDENTICAL input (the noise suppressor's output of the exact build), and after every packet the decisions of
NonLinearProcessing are compared (orc_aec_probe): near-end state, echo state, divergence state, the delay partition, the
suppression minimum's bookkeeping.  Prints one JSON object; `python tests/evidence_fma_contraction.py > profiles/r04/aec_contraction_cpu.json`.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loader as L  # noqa: E402
from wmix_amd import synth  # noqa: E402

INTS = ("stNearState", "echoState", "divergeState", "delayIdx", "hNlNewMin", "hNlMinCtr", "noise_ctr", "system_delay")
FLTS = ("hNlFbMin", "hNlFbLocalMin", "hNlXdAvgMin", "overDrive", "overDriveSm", "sum_sd", "sum_se")


def _bind(lib):
    i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
    lib.orc_aec_init.restype = C.c_void_p
    lib.orc_aec_init.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.orc_aec_process2.restype = C.c_int
    lib.orc_aec_process2.argtypes = [C.c_void_p, i16p, i16p, i16p, C.c_int, C.c_int]
    lib.orc_aec_probe.restype = None
    lib.orc_aec_probe.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.int32), np.ctypeslib.ndpointer(np.float32)]
    lib.orc_aec_release.restype = None
    lib.orc_aec_release.argtypes = [C.c_void_p]
    return lib


def main(n_streams=64, n=3000, freq=16000):
    pkt = freq // 100
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port", "fma"])
    exact = _bind(C.CDLL(os.path.join(ROOT, "oracle", "build", "liboracle.so")))
    fused = _bind(C.CDLL(os.path.join(ROOT, "oracle", "build", "liboracle_fma.so")))
    far = synth.far_end(8001, n, pkt)                      # the recipe of tests/test_aec_gpu.py::test_chain_parity_gate_3000_frames
    near = synth.near_end(8100, 256, n, pkt, far=far)
    pick = np.random.default_rng(8).choice(256, 64, replace=False)[:n_streams]
    rows, worst = [], 0
    for s in pick:
        x = L.run_ns(L.port(), 1, freq, near[s], pkt, prefix="orc")  # the AEC's input: identical for both builds
        a, b = exact.orc_aec_init(1, freq, 10), fused.orc_aec_init(1, freq, 10)
        ia, ib = np.zeros(8, np.int32), np.zeros(8, np.int32)
        fa, fb = np.zeros(7, np.float32), np.zeros(7, np.float32)
        oa, ob = np.zeros(pkt, np.int16), np.zeros(pkt, np.int16)
        first_lsb, first_big, first_flip, flip_what, max_lsb, n_off = None, None, None, None, 0, 0
        at_flip = None
        max_early, max_late, first_big_late, nan_like = 0, 0, None, 0  # "early": the 50 packets behind the start-up pass-through
        for p in range(n):
            f = np.ascontiguousarray(far[p * pkt:(p + 1) * pkt])
            q = np.ascontiguousarray(x[p * pkt:(p + 1) * pkt])
            assert exact.orc_aec_process2(a, f, q, oa, pkt, 0) == 0 and fused.orc_aec_process2(b, f, q, ob, pkt, 0) == 0
            d = np.abs(oa.astype(np.int32) - ob.astype(np.int32))
            m = int(d.max())
            n_off += int((d > 0).sum())
            max_lsb = max(max_lsb, m)
            if m >= 1 and first_lsb is None:
                first_lsb = p
            if m > 1 and first_big is None:
                first_big = p
            if p < 56:
                max_early = max(max_early, m)
                # the reference's first blocks produce NaNs (powf of a slightly negative suppression gain, DESIGN_HISTORY section 2), which
                # the int16 conversion turns into 0: a sample that is 0 in one build and far from 0 in the other is that
                nan_like += int((((oa == 0) != (ob == 0)) & (d > 1)).sum())
            else:
                max_late = max(max_late, m)
                if m > 1 and first_big_late is None:
                    first_big_late = p
            exact.orc_aec_probe(a, ia, fa)
            fused.orc_aec_probe(b, ib, fb)
            if first_flip is None and not np.array_equal(ia[:6], ib[:6]):
                first_flip = p
                flip_what = [INTS[k] for k in range(6) if ia[k] != ib[k]]
                at_flip = {"exact": dict(zip(INTS, ia.tolist())) | dict(zip(FLTS, [float(v) for v in fa])),
                           "fused": dict(zip(INTS, ib.tolist())) | dict(zip(FLTS, [float(v) for v in fb]))}
        exact.orc_aec_release(a)
        fused.orc_aec_release(b)
        worst = max(worst, max_lsb)
        rows.append({"stream": int(s), "max_lsb": max_lsb, "max_lsb_packets_0_55": max_early, "max_lsb_packets_56_on": max_late,
                     "first_packet_off_by_more_from_56_on": first_big_late, "zero_vs_nonzero_samples_in_packets_0_55": nan_like,
                     "samples_differing": n_off, "first_packet_off_by_one": first_lsb,
                     "first_packet_off_by_more": first_big, "first_packet_with_a_flipped_decision": first_flip,
                     "flipped": flip_what, "state_at_flip": at_flip})
    big = [r for r in rows if r["max_lsb"] > 1]
    out = {"what": "float AEC restatement, exact build (-ffp-contract=off) against -mfma -ffp-contract=fast, same input, packet by packet",
           "streams": len(rows), "packets_per_stream": n, "rate": freq,
           "streams_within_1_lsb": len(rows) - len(big), "streams_beyond_1_lsb": len(big), "worst_lsb": worst,
           "streams_beyond_1_lsb_after_the_first_50_active_packets": sum(1 for r in rows if r["max_lsb_packets_56_on"] > 1),
           "worst_lsb_after_the_first_50_active_packets": max(r["max_lsb_packets_56_on"] for r in rows),
           "every_excursion_beyond_1_lsb_starts_with_a_flipped_decision":
               all(r["first_packet_with_a_flipped_decision"] is not None and r["first_packet_with_a_flipped_decision"] <= r["first_packet_off_by_more"]
                   for r in big),
           "flipped_decisions_histogram": {k: sum(1 for r in big if r["flipped"] and k in r["flipped"]) for k in INTS[:6]},
           "streams_beyond": big, "streams_within": [{k: r[k] for k in ("stream", "max_lsb", "samples_differing")} for r in rows if r["max_lsb"] <= 1]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
