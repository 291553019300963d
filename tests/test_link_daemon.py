"""Drop-in boundary, SURVEY.md section 8b: the unchanged daemon objects of the reference link against libwmix_amd.so and
every boundary symbol -- the src/webrtc.h and src/g711codec.h groups dynamically, the src/wmix.h group through the
weaken + daemon_shim.o recipe of INTEGRATION.md section 2 -- resolves to our library.  Build container only: the
reference tree does not exist on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs /root/reference (build container only)")
def test_daemon_links_against_libwmix_amd(wmx):
    r = subprocess.run(["bash", os.path.join(ROOT, "tools_dev", "link_daemon.sh")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    for s in ("wmix_load_data", "wmix_pcm_zoom", "wmix_len_of_out", "wmix_len_of_in"):
        assert "%s <- daemon_shim.o (nm: T)" % s in out
    for s in ("vad_init", "aec_process2", "ns_process", "agc_addition", "PCM2G711a", "G711a2PCM"):
        assert "%s <- libwmix_amd.so (dynamic: U)" % s in out
    assert out.strip().endswith("resolves to libwmix_amd")


def test_shim_compiles_as_plain_c(tmp_path):
    """daemon_shim.c is what the maintainer's C compiler sees: C (gnu99, the default dialect the reference builds with), no HIP, only include/wmix_compat.h."""
    o = tmp_path / "shim.o"
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Werror", "-c", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "wmix_amd", "csrc", "daemon_shim.c"), "-o", str(o)])
    syms = subprocess.check_output(["nm", str(o)], text=True)
    for s in ("T wmix_load_data", "T wmix_pcm_zoom", "T wmix_len_of_out", "T wmix_len_of_in", "U wmx_compat_load_data"):
        assert s in syms
