#!/bin/bash
# link_daemon.sh -- build container only (needs /root/reference): link the UNCHANGED wmix daemon objects against
# libwmix_amd.so and show, in the linker's own words, which definition of every boundary symbol wins.
#
#   * src/{wmix,wmixTask,wmixMem,rtp,wav,delay}.c compiled from where they lie, with the reference Makefile's flags
#     (no -O; MP3/AAC/SPEEX off because their third-party libs are not built here), WITHOUT src/webrtc.c and
#     src/g711codec.c -- that is INTEGRATION.md section 2;
#   * the wmix.h group lives in wmix.o itself: the four symbols are weakened with objcopy and
#     wmix_amd/csrc/daemon_shim.c supplies the strong definitions that forward into libwmix_amd.so;
#   * the platform HAL (plat_*: ALSA, not in this image) and the codec wrappers stay undefined -- nothing is stubbed,
#     the executable is linked with --unresolved-symbols=ignore-all and never run.
#
# Nothing from the reference is copied into the repo: objects, map and binary live in a temp dir that is deleted.
# Output: one line per boundary symbol "sym <- provider", then OK / FAIL.  Exit 0 only if every symbol is ours.
set -euo pipefail
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LIB=$ROOT/wmix_amd/libwmix_amd.so
[ -d "$REF/src" ] || { echo "no $REF: nothing to link"; exit 3; }
[ -f "$LIB" ] || { echo "libwmix_amd.so not built"; exit 2; }
T=$(mktemp -d /tmp/wmixlink.XXXXXX)
trap 'rm -rf "$T"' EXIT
DEF="-DMAKE_MP3=0 -DMAKE_AAC=0 -DMAKE_SPEEX=0 -DMAKE_SPEEX_BETA3=0 -DMAKE_MATH_FFT=0"
for f in wmix wmixTask wmixMem rtp wav delay; do
    gcc -w -c $DEF -I"$REF/src" -I"$REF/platform/alsa" "$REF/src/$f.c" -o "$T/$f.o"
done
GROUP="wmix_load_data wmix_pcm_zoom wmix_len_of_out wmix_len_of_in"
DSP="vad_init vad_process vad_release aec_init aec_process2 aec_release ns_init ns_process ns_release agc_init agc_process agc_addition agc_release PCM2G711a G711a2PCM"
W=""; for s in $GROUP; do W="$W --weaken-symbol=$s"; done
objcopy $W "$T/wmix.o"
gcc -c -I"$ROOT/include" "$ROOT/wmix_amd/csrc/daemon_shim.c" -o "$T/daemon_shim.o"
Y=""; for s in $GROUP $DSP; do Y="$Y -Wl,-y,$s"; done
gcc -no-pie -o "$T/wmix_linked" "$T"/{wmix,wmixTask,wmixMem,rtp,wav,delay,daemon_shim}.o \
    -L"$ROOT/wmix_amd" -lwmix_amd -Wl,-rpath,"$ROOT/wmix_amd" -Wl,--unresolved-symbols=ignore-all \
    -Wl,-Map="$T/link.map" $Y -lm -lpthread >"$T/trace.txt" 2>&1 || { cat "$T/trace.txt"; echo FAIL: link; exit 1; }
rc=0
for s in $GROUP; do
    # the executable's own definition must be the shim's: the map lists a symbol under the input section that supplied it
    prov=$(awk -v s="$s" '/^ \.text/ {obj=$NF} $2==s && NF==2 {print obj}' "$T/link.map" | head -1)
    addr=$(nm "$T/wmix_linked" | awk -v s="$s" '$3==s {print $2}')
    echo "$s <- $(basename "${prov:-?}") (nm: ${addr:-?})"
    case "$prov" in *daemon_shim.o) [ "$addr" = T ] || rc=1 ;; *) rc=1 ;; esac
done
for s in $DSP; do
    def=$(grep "definition of $s\$" "$T/trace.txt" | sed -e 's/: definition.*//' -e 's/^.*ld: //' | xargs -n1 basename | tr '\n' ' ')
    und=$(nm -D "$T/wmix_linked" | awk -v s="$s" '$2==s {print $1}')
    echo "$s <- ${def:-?}(dynamic: ${und:-?})"
    [ "$def" = "libwmix_amd.so " ] && [ "$und" = U ] || rc=1
done
readelf -d "$T/wmix_linked" | grep -q 'NEEDED.*libwmix_amd.so' || { echo "libwmix_amd.so is not DT_NEEDED"; rc=1; }
# the collision-free targets of the shim come from the library too
for s in wmx_compat_load_data wmx_compat_pcm_zoom wmx_compat_len_of_out wmx_compat_len_of_in; do
    nm -D "$T/wmix_linked" | grep -q " U $s\$" || { echo "$s not bound to the library"; rc=1; }
done
# the two call sites inside wmix.o itself (src/wmix.c:718,736) must land on the shim's addresses
for s in wmix_load_data wmix_pcm_zoom; do
    addr=$(nm "$T/wmix_linked" | awk -v s="$s" '$3==s {print $1}' | sed 's/^0*//')
    calls=$(objdump -d "$T/wmix_linked" | grep -c "call.* $addr <$s>" || true)
    echo "calls to $s bound to the shim at 0x$addr: $calls"
    [ "$calls" -ge 1 ] || rc=1
done
[ $rc = 0 ] && echo "OK: every boundary symbol of src/webrtc.h, src/g711codec.h and src/wmix.h resolves to libwmix_amd" || echo "FAIL"
exit $rc
