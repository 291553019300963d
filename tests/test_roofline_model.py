"""The issue-side figures of bench.py's roofline object are made from files committed under profiles/rNN/ (the PMC instruction
counts of the profiled command, the measured instruction prices, the kernel's class histogram).  No GPU needed to check that
they fit together: the lower bound is a bound, the ALU-busy estimate lies between it and the launch, the scalar side is there."""
import importlib.util
import json
import os

import pytest

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("tag,kernel,n_frames", [("chain", "aec_near_kernel<2>", 65536), ("ns_aec_8k", "aec_near_kernel<1>", 131072),
                                                 ("nsx", "nsx_kernel<256, 1>", 65536), ("aecm", "aecm_near_kernel", 65536)])
def test_issue_accounting_from_committed_profiles(tag, kernel, n_frames):
    b = _bench()
    r = b._pmc_issue(kernel, n_frames, tag, 1.0)
    assert r is not None and r["source"].startswith("profiles/r")
    rnd = r["source"].split("/")[1]  # the newest round that profiled this workload at this size
    line = json.load(open(os.path.join(ROOT, "profiles", rnd, tag + "_bench_line_under_rocprof.json")))
    launch_ms = line["roofline"]["avg_launch_ms"]
    r = b._pmc_issue(kernel, n_frames, tag, launch_ms)
    assert 0.1 < r["lower_bound_frac"] < r["mix_estimate_frac"] <= 1.0, r
    assert r["priced_at_waves_per_simd"] <= r["waves_per_simd"] and r["priced_at_waves_per_simd"] <= 4  # only checked columns of the table
    assert 0 < r["scalar_ceiling_frac"] < 1 and r["salu_insts_per_frame"] > 0
    traffic, src = b._pmc_traffic(kernel, n_frames, tag)
    assert src == "profiles/%s/%s_hbm_pmc.json" % (rnd, tag) and 0.5 < traffic / line["roofline"]["algorithmic_bytes_per_launch"] < 1.3


def test_price_table_has_its_own_residency_check():
    t = json.load(open(os.path.join(ROOT, "profiles", "r03", "issue_costs.json")))
    ws = t["waves_per_simd"]
    add = t["classes"]["v_add_f32"]
    i4 = ws.index(4)
    assert add["resident_waves_per_simd"][i4] > 3.8 and 1.9 < add["cycles"][i4] < 2.3  # the guide's 2 cycles, at four waves per SIMD
    assert 3.8 < t["classes"]["v_pk_add_f32"]["cycles"][i4] < 4.2 and 3.3 < t["classes"]["v_fma_f32"]["cycles"][i4] < 3.9
