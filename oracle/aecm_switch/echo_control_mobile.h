/* oracle/aecm_switch/echo_control_mobile.h -- TEST INFRASTRUCTURE ONLY, used by `make -C oracle ref` alone.
 *
 * The reference selects AECM instead of the float AEC by a hand edit: un-commenting `#undef MAKE_WEBRTC_AEC` between the
 * two #include lines and the `#ifdef MAKE_WEBRTC_AEC` that follows them (src/webrtc.c:168-191).  That cannot be said on a
 * compiler command line (the macro must be non-zero for the enclosing `#if (MAKE_WEBRTC_AEC)` and undefined two lines
 * later), and the reference tree is read-only.  This two-line shim sits IN FRONT of the real echo_control_mobile.h on the
 * include path of the one extra compilation of src/webrtc.c: it includes the real header, then performs exactly that
 * edit.  No reference text is reproduced here and nothing is stubbed: every declaration comes from the vendored header.
 */
#include_next "echo_control_mobile.h"
#undef MAKE_WEBRTC_AEC
