"""The library says which build it is (round-4 VERDICT "weak" 8): the product build is "default"; a developer variant is refused by the
Python mirror unless asked for, and the wrong-result timing switches do not compile without -DWMX_TIMING_ONLY_BUILD."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "wmix_amd", "csrc")


def test_product_build_is_default():
    from wmix_amd import _lib
    assert _lib.build_info() == "default"
    assert "wmx_build_info" in _lib.declared_symbols()


@pytest.mark.parametrize("flag,src", [("-DWMX_AEC_EXP=2", "aec.hip"), ("-DWMX_AEC_EXP_BARRIERS", "aec.hip"), ("-DWMX_NS_EXP=1", "ns.hip")])
def test_wrong_result_switches_do_not_compile_silently(flag, src):
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", flag, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "WMX_TIMING_ONLY_BUILD" in r.stderr


def test_a_variant_build_is_refused_unless_asked_for(tmp_path):
    """wmx_core.hip alone, built with a developer flag, linked into a stub library: loading it through wmix_amd._lib fails loudly."""
    obj = tmp_path / "core.o"
    so = tmp_path / "libvariant.so"
    flags = ["-O1", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-DWMX_AEC_WAVES=5", "-DWMX_BUILD_EXTRA=\"-DWMX_AEC_WAVES=5\""]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CSRC, "wmx_core.hip"), "-o", str(obj)], timeout=600)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(so), str(obj)], timeout=600)
    code = ("import ctypes, sys\n"
            "L = ctypes.CDLL(sys.argv[1]); L.wmx_build_info.restype = ctypes.c_char_p\n"
            "print(L.wmx_build_info().decode())\n")
    r = subprocess.run([sys.executable, "-c", code, str(so)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "-DWMX_AEC_WAVES=5", (r.stdout, r.stderr)
    env = dict(os.environ, WMIX_AMD_LIB=str(so), PYTHONPATH=ROOT)
    env.pop("WMIX_AMD_ALLOW_VARIANT_BUILD", None)
    r = subprocess.run([sys.executable, "-c", "from wmix_amd import _lib; _lib.lib()"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "not the product build" in r.stderr and "-DWMX_AEC_WAVES=5" in r.stderr
