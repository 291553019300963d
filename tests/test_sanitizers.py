"""Sanitizers on everything that runs on the host: AddressSanitizer +
UndefinedBehaviorSanitizer with -fno-sanitize-recover, so a finding kills the child process and fails the test.
GPU AddressSanitizer is not available on this pool; the kernels are covered by the parity tests instead."""
import glob
import os
import subprocess
import sys

from conftest import ROOT


def test_host_control_planes_under_asan_ubsan():
    """aec_ctl.h / aecm_ctl.h / agc_gain_table.h / mix_sched.h -- the host code of libwmix_amd.so that decides where data
    goes -- compiled without HIP and driven over their argument ranges (tools_dev/san/host_ctl_san.cpp)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools_dev", "san")])
    r = subprocess.run([os.path.join(ROOT, "tools_dev", "san", "host_ctl_san")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "clean under ASan + UBSan" in r.stdout


def test_oracle_tests_against_the_sanitized_restatement():
    """Every oracle test (restatement vs the real reference, goldens, upstream known answers) once more in a child
    interpreter that loads oracle/build/liboracle_san.so (WMIX_ORACLE_SAN=1, ASan runtime preloaded)."""
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.exists(asan), "gcc's libasan.so not found"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "test_*_oracle.py"))) + [os.path.join(ROOT, "tests", "test_pkgfifo.py")]
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", WMIX_ORACLE_SAN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-s", "-p", "no:cacheprovider", "-m", "not gpu"] + files,
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in tail and "AddressSanitizer" not in tail
