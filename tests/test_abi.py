"""CPU tests of the drop-in boundary: libwmix_amd.so loads and exports every
symbol include/*.h declares (no compute calls; no GPU needed)."""
import ctypes as C

from wmix_amd import _lib


def test_library_loads_and_exports_every_declared_symbol(wmx):
    names = _lib.declared_symbols()
    assert "wmx_g711_encode" in names and "PCM2G711a" in names
    missing = [n for n in names if not hasattr(wmx, n)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_version_and_error_string(wmx):
    assert wmx.wmx_version() >= 100
    assert isinstance(wmx.wmx_last_error(), bytes)


def test_bad_arguments_are_rejected_without_touching_the_gpu(wmx):
    # invalid law -> WMX_EINVAL before any HIP call
    assert wmx.wmx_g711_encode(7, None, None, 16, None) == -10001
    assert b"law" in wmx.wmx_last_error()
    assert wmx.wmx_g711_decode(-1, None, None, 16, None) == -10001
    # n == 0 is a no-op success like the reference loops (src/g711codec.c:194-216)
    assert wmx.wmx_g711_encode(0, None, None, 0, None) == 0
    # reference null check: -1 only when in, out and len are all null/0 (src/g711codec.c:230)
    assert wmx.PCM2G711a(None, None, 0, 0) == -1
    assert wmx.G711u2PCM(None, None, 0, 0) == -1


def test_headers_stand_alone_under_strict_c99(tmp_path):
    """include/*.h must compile on their own with -std=c99 -pedantic (no _DEFAULT_SOURCE needed)."""
    import os
    import subprocess
    for h in ("wmix_compat.h", "wmix_amd.h"):
        src = tmp_path / ("t_" + h.replace(".h", ".c"))
        src.write_text('#include "%s"\nint main(void) { return 0; }\n' % h)
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + _lib.INCLUDE_DIR, "-c", str(src), "-o",
                               str(tmp_path / "t.o")])
