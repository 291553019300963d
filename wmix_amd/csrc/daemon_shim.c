/* daemon_shim.c -- the one object a wmix maintainer adds to the daemon's link so that the wmix.h group
 * (wmix_load_data, wmix_pcm_zoom, wmix_len_of_out, wmix_len_of_in; src/wmix.h:40-49,113-127) runs in libwmix_amd.so.
 *
 * Those four are defined in src/wmix.c, the same translation unit as main(), so dropping an object from the link line
 * (what INTEGRATION.md does for src/webrtc.c and src/g711codec.c) cannot remove them, and a definition in a shared
 * library never overrides one in the executable.  The build-only recipe (no source change in the daemon):
 *
 *     objcopy --weaken-symbol=wmix_load_data --weaken-symbol=wmix_pcm_zoom \
 *             --weaken-symbol=wmix_len_of_out --weaken-symbol=wmix_len_of_in src/wmix.o
 *     gcc -c $(WMIX_AMD)/wmix_amd/csrc/daemon_shim.c -I$(WMIX_AMD)/include -o daemon_shim.o
 *     ... link daemon_shim.o and -lwmix_amd with the other objects ...
 *
 * The strong definitions below then win at static link time for every caller, including the two call sites inside
 * wmix.c itself (src/wmix.c:718,736; the reference is built without -O, so they are calls by symbol).  Plain C, no HIP:
 * it only forwards to the collision-free names exported by libwmix_amd.so.  tools_dev/link_daemon.sh performs
 * exactly this against /root/reference and shows which definition the linker picked.
 */
#include "wmix_compat.h"

WMix_Point wmix_load_data(WMix_Struct_Head *wmix, WMix_Point src, uint32_t srcU8Len, uint16_t freq, uint8_t channels,
                          uint8_t sample, WMix_Point head, uint8_t reduce, uint32_t *tick)
{
    return wmx_compat_load_data(wmix, src, srcU8Len, freq, channels, sample, head, reduce, tick);
}

uint32_t wmix_len_of_out(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq)
{
    return wmx_compat_len_of_out(inChn, inFreq, inLen, outChn, outFreq);
}

uint32_t wmix_len_of_in(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t outLen)
{
    return wmx_compat_len_of_in(inChn, inFreq, outChn, outFreq, outLen);
}

uint32_t wmix_pcm_zoom(uint8_t inChn, uint16_t inFreq, uint8_t *in, uint32_t inLen, uint8_t outChn, uint16_t outFreq,
                       uint8_t *out)
{
    return wmx_compat_pcm_zoom(inChn, inFreq, in, inLen, outChn, outFreq, out);
}
