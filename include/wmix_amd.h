/* wmix_amd.h -- C ABI of libwmix_amd.so: the MI355X (gfx950) implementation of
 * wmix's per-frame DSP hot path.
 *
 * Two layers are exported:
 *
 *  (1) the REFERENCE'S OWN signatures, unchanged, so the wmix daemon links
 *      against libwmix_amd.so instead of src/webrtc.c + libwebrtc{vad,aec,ns,agc}
 *      + src/g711codec.c (+ the two arithmetic functions of src/wmix.c).  They take
 *      HOST pointers exactly like the reference and are thin adapters over a
 *      batch of one stream (H2D copy, one launch, D2H copy).  Declared in
 *      include/wmix_compat.h.
 *
 *  (2) the BATCHED entry points below (prefix wmx_): many independent 10 ms
 *      streams per launch, PCM and state resident in HBM.  Plain pointers and
 *      sizes only; `stream` is a hipStream_t passed as void* (NULL = default
 *      stream).  All d_* pointers are DEVICE pointers.  Every function returns 0
 *      on success, a negative value on failure (WMX_EHIP_BASE - hipError_t for HIP
 *      errors, WMX_E* otherwise); wmx_last_error() describes the last failure on the
 *      calling thread.  There is NO CPU fallback: without a usable HIP device the
 *      calls fail.
 */
#ifndef WMIX_AMD_H
#define WMIX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WMX_EINVAL (-10001) /* bad argument (unsupported freq / chn / size) */
#define WMX_ENODEV (-10002) /* no HIP device */
#define WMX_ESTATE (-10003) /* wrong handle kind */
/* A failing HIP runtime call returns WMX_EHIP_BASE - (int)hipError_t (hipErrorOutOfMemory = 2 -> -11002): outside the
 * reference's own return values (0 / -1), which some entry points hand through (wmx_aec_run: -1 = the wrapper stopped
 * at a delay outside [0, 500]). */
#define WMX_EHIP_BASE (-11000)
#define WMX_IS_HIP_ERROR(rc) ((rc) <= WMX_EHIP_BASE && (rc) > WMX_EHIP_BASE - 10000)

const char *wmx_last_error(void);
int wmx_device_count(void);
/* Device affinity: every wmx_* handle records the HIP device that is current on the creating thread and every entry
 * point that takes the handle runs on that device (and restores the caller's), so a host with one thread per GPU needs
 * no thread-local hipSetDevice discipline.  Device pointers passed to a call must live on
 * the handle's device.  Works for any wmx_* handle type; WMX_EINVAL for NULL. */
int wmx_handle_device(const void *handle);
/* library/ABI version: major*10000 + minor*100 + patch */
int wmx_version(void);
/* "default" for the product build; else the developer flags the library was built with (prefixed "TIMING-ONLY (wrong results): "
 * when one of them is a timing experiment's switch).  The Python mirror, build(), smoke() and bench.py refuse anything but "default"
 * unless WMIX_AMD_ALLOW_VARIANT_BUILD=1. */
const char *wmx_build_info(void);

/* ------------------------------------------------------------------ per-stream lifetime inside a batch
 * The reference creates each handle lazily in the record heartbeat and releases it when its switch drops or recording
 * idles (src/wmix.c:565-600, 617-618, 635-636, 683-684, 702-703, 783-813; *_init / *_release in src/webrtc.c:40-82,
 * 153-164, 217-274, 485-505, 560-602, 646-661, 694-753, 841-860): the streams of a batch join, leave and restart on their
 * own.  Every stateful module therefore has
 *   wmx_<m>_reset_streams(h, idx, n, stream)  <m>_release + <m>_init for the n streams listed in the HOST array idx (a
 *                                             refill kernel ordered on `stream` like the process calls around it);
 *   wmx_<m>_set_active(h, mask, stream)       mask: HOST array of n_streams bytes; 0 = the stream is not called -- state
 *                                             and PCM rows untouched, like a handle nobody calls; NULL = all active.  The
 *                                             mask stays in force until the next set_active.
 * The AEC / AECM, whose control plane is shared by the streams that were started together, add COHORTS, see there.
 * Bad index -> WMX_EINVAL, nothing reset. */

/* Stream migration between batches / GPUs (a stream that outlives its batch, or that moves to another device for balance):
 *   wmx_<m>_stream_state_bytes(h)            size of one stream's state blob for this handle's format
 *   wmx_<m>_export_stream(h, i, blob)        the complete state of stream i into HOST memory (blocking: the device is drained)
 *   wmx_<m>_import_stream(h, i, blob[, c])   the reverse, into any handle of the same module and format (WMX_ESTATE otherwise);
 *                                            for the AEC / AECM `c` >= 0 also makes the stream a member of cohort c
 *   wmx_{aec,aecm}_cohort_state_bytes / _export_cohort / _import_cohort   the shared part: control plane + far-end history
 * A stream imported together with its cohort continues bit for bit as if it had never moved (tests/test_lifetime_gpu.py). */

/* ------------------------------------------------------------------ G.711
 * Replaces g711{a,u}_encode / g711{a,u}_decode (src/g711codec.h:30-34,
 * src/g711codec.c:82-216) for device-resident buffers.  law: 0 = A-law, 1 = mu-law.
 * Bit-exact with the reference including its 16-bit-domain segment table and the
 * negative A-law path. */
#define WMX_LAW_A 0
#define WMX_LAW_U 1
int wmx_g711_encode(int law, const int16_t *d_pcm, uint8_t *d_code, size_t n_samples, void *stream);
int wmx_g711_decode(int law, const uint8_t *d_code, int16_t *d_pcm, size_t n_codes, void *stream);

/* ------------------------------------------------------------------ NS (float noise suppressor)
 * Batched form of ns_init / ns_process / ns_release (src/webrtc.h:47-51, src/webrtc.c:560-661):
 * n_streams independent streams, each with the state WebRtcNs_Create/Init/set_policy(2) would
 * give it.  freq in {8000,16000,32000}, chn in {1,2}; anything else -> WMX_EINVAL (ns_init
 * returns NULL there; 24000 passes the wrapper's `freq % 8000` test but WebRtcNs_Init refuses it, and for VAD / AGC
 * the reference then hands out a handle whose every process call fails -- here creation fails for all four).  A packet is 10 ms = freq/100 frames of chn interleaved int16.
 *
 * wmx_ns_process runs n_packets consecutive packets of every stream in one launch.  Packet p of
 * stream s starts at d_in + s*stream_stride + p*packet_stride (strides in int16 elements), same
 * for d_out; d_out may alias d_in (the daemon always processes in place).  Reference quirks are
 * kept: the right channel of a 2-channel stream is treated as a high band, and at 32 kHz only the
 * first 160 frames of each packet are processed, the rest of the output packet is zero
 * (SURVEY.md section 0 quirks 2-3).
 *
 * Every spectral / time sum is added in the reference's index order: bit-exact with the CPU path.  (The faster re-associated mode of
 * rounds 1-5, wmx_ns_set_ordered(h, 0), was outside north_star's +-1 LSB -- up to 13 LSB on a few samples -- and is gone.) */
typedef struct wmx_ns wmx_ns;
int wmx_ns_create(wmx_ns **out, int n_streams, int chn, int freq);
int wmx_ns_destroy(wmx_ns *h);
int wmx_ns_packet_samples(const wmx_ns *h); /* int16 elements per packet = freq/100*chn */
int wmx_ns_process(wmx_ns *h, const int16_t *d_in, int16_t *d_out, int n_packets, long stream_stride,
                   long packet_stride, void *stream);
/* debugging / tests: copy one stream's state block (layout: wmix_amd/csrc/ns_layout.h) and its
 * 3 x 1000 histogram counters to host memory (either pointer may be NULL). */
int wmx_ns_state_words(const wmx_ns *h);
int wmx_ns_export_state(const wmx_ns *h, int stream_index, float *host_words, unsigned short *host_hist);
int wmx_ns_reset_streams(wmx_ns *h, const int32_t *idx, int n, void *stream);
int wmx_ns_set_active(wmx_ns *h, const uint8_t *host_mask, void *stream);
int wmx_ns_stream_state_bytes(const wmx_ns *h);
int wmx_ns_export_stream(wmx_ns *h, int stream_index, void *host_blob);
int wmx_ns_import_stream(wmx_ns *h, int stream_index, const void *host_blob);

/* ------------------------------------------------------------------ NSX (fixed-point noise suppressor)
 * Batched form of the SAME three wrapper functions when the reference is built with its MAKE_WEBRTC_NSX switch
 * (src/webrtc.c:512-521: ns_init / ns_process / ns_release over WebRtcNsx_Create / Init / set_policy(2) / Process,
 * W:modules/audio_processing/ns/nsx_core.c).  Same packet rule, strides, aliasing and quirks as wmx_ns_* (right channel =
 * high band, second half of a 32 kHz packet zero).  Integer path: bit-exact.  The legacy ns_init picks this
 * implementation when the environment has WMIX_AMD_NSX=1 (the build-time macro of the reference becomes a run-time
 * switch). */
typedef struct wmx_nsx wmx_nsx;
int wmx_nsx_create(wmx_nsx **out, int n_streams, int chn, int freq);
int wmx_nsx_destroy(wmx_nsx *h);
int wmx_nsx_packet_samples(const wmx_nsx *h);
int wmx_nsx_state_bytes(const wmx_nsx *h); /* per-stream state block in HBM (the 3 x 1000 int16 histograms come on top) */
int wmx_nsx_process(wmx_nsx *h, const int16_t *d_in, int16_t *d_out, int n_packets, long stream_stride,
                    long packet_stride, void *stream);
int wmx_nsx_reset_streams(wmx_nsx *h, const int32_t *idx, int n, void *stream);
int wmx_nsx_set_active(wmx_nsx *h, const uint8_t *host_mask, void *stream);
int wmx_nsx_stream_state_bytes(const wmx_nsx *h);
int wmx_nsx_export_stream(wmx_nsx *h, int stream_index, void *host_blob);
int wmx_nsx_import_stream(wmx_nsx *h, int stream_index, const void *host_blob);

/* ------------------------------------------------------------------ VAD (voice-activity gate)
 * Batched form of vad_init / vad_process / vad_release (src/webrtc.h:32-36, src/webrtc.c:40-164):
 * WebRtcVad mode 3 decides speech/no-speech per packet; a per-stream `reduce` in [0,4] moves one
 * step per decision and the packet is shifted right by it, in place.  Packet = intervalMs worth of
 * frames (20 ms only when freq <= 16000 and interval_ms % 20 == 0, else 10 ms).
 *
 * One "call" = one vad_process() invocation covering packets_per_call packets; the reference
 * analyses and attenuates only packet 0 of a call and so do we.  Call c of
 * stream s starts at d_pcm + s*stream_stride + c*call_stride (int16 elements).  Integer path:
 * bit-exact. */
typedef struct wmx_vad wmx_vad;
int wmx_vad_create(wmx_vad **out, int n_streams, int chn, int freq, int interval_ms);
int wmx_vad_destroy(wmx_vad *h);
int wmx_vad_packet_samples(const wmx_vad *h); /* int16 elements per packet = freq/1000*intervalMs*chn */
int wmx_vad_process(wmx_vad *h, int16_t *d_pcm, int packets_per_call, int n_calls, long stream_stride,
                    long call_stride, void *stream);
int wmx_vad_reset_streams(wmx_vad *h, const int32_t *idx, int n, void *stream);
int wmx_vad_set_active(wmx_vad *h, const uint8_t *host_mask, void *stream);
int wmx_vad_stream_state_bytes(const wmx_vad *h);
int wmx_vad_export_stream(wmx_vad *h, int stream_index, void *host_blob);
int wmx_vad_import_stream(wmx_vad *h, int stream_index, const void *host_blob);

/* ------------------------------------------------------------------ AGC (legacy fixed-point, adaptive digital)
 * Batched form of agc_init / agc_process / agc_addition / agc_release (src/webrtc.h:55-60,
 * src/webrtc.c:694-860): target 0 dBFS, compression gain `value` dB, limiter off.  Channels are
 * averaged to mono, processed as one band and duplicated back.  Packet = 10 ms (5 ms at 32 kHz,
 * SURVEY.md section 0 quirk 4).  `value` outside the reference's gain-table range makes create/set_gain
 * fail with WMX_EINVAL (agc_init returns NULL there).  Integer path: bit-exact.
 *
 * The compression gain is PER STREAM, as it is per handle in the reference (agc_init's `value`, agc_addition(fp, value);
 * the daemon: src/wmix.c:684, 1068-1070): wmx_agc_create / wmx_agc_set_gain give every stream of the batch one value;
 * wmx_agc_set_gain_streams is agc_addition for the listed streams (state untouched, new gain curve from the next packet on),
 * wmx_agc_reset_streams_gain is agc_release + agc_init(.., value, ..) for them; wmx_agc_reset_streams re-creates a stream
 * with the gain it has.  One 32-entry table per distinct value in use; a batch whose streams all share one value runs
 * the same kernels as before.  The value travels with wmx_agc_export_stream / _import_stream. */
typedef struct wmx_agc wmx_agc;
int wmx_agc_create(wmx_agc **out, int n_streams, int chn, int freq, int interval_ms, int value);
int wmx_agc_destroy(wmx_agc *h);
int wmx_agc_set_gain(wmx_agc *h, int value); /* agc_addition for every stream of the batch (blocking) */
/* idx: HOST array of n stream indices; ordered on `stream` like the process calls around it.  A value the reference's
 * WebRtcAgc_set_config refuses -> WMX_EINVAL, nothing changed. */
int wmx_agc_set_gain_streams(wmx_agc *h, const int32_t *idx, int n, int value, void *stream);
int wmx_agc_reset_streams_gain(wmx_agc *h, const int32_t *idx, int n, int value, void *stream);
int wmx_agc_stream_gain(const wmx_agc *h, int stream_index); /* the compression gain stream i runs with */
int wmx_agc_packet_samples(const wmx_agc *h);
int wmx_agc_gain_table(const wmx_agc *h, int32_t *host_table32); /* the 32 Q16 gains of the batch's own value (tests) */
int wmx_agc_process(wmx_agc *h, const int16_t *d_in, int16_t *d_out, int n_packets, long stream_stride,
                    long packet_stride, void *stream);
int wmx_agc_reset_streams(wmx_agc *h, const int32_t *idx, int n, void *stream); /* each stream keeps its gain */
int wmx_agc_set_active(wmx_agc *h, const uint8_t *host_mask, void *stream);
int wmx_agc_stream_state_bytes(const wmx_agc *h);
int wmx_agc_export_stream(wmx_agc *h, int stream_index, void *host_blob);
int wmx_agc_import_stream(wmx_agc *h, int stream_index, const void *host_blob);

/* ------------------------------------------------------------------ AEC (float echo canceller)
 * Batched form of aec_init / aec_setFrameFar / aec_process / aec_process2 / aec_release
 * (src/webrtc.h:40-45, src/webrtc.c:217-505): n_streams near-end streams cancelled against ONE
 * shared far-end reference (BASELINE.json configs[2..3]; a batch of one stream is the reference's
 * per-handle case).  freq in {8000,16000}; packet = 10 ms (20 ms only at 8000 Hz with
 * interval_ms % 20 == 0).  Only channel 0 of far and near is used; the output is duplicated to
 * every channel.
 *
 * wmx_aec_run processes n_packets consecutive packets: mode bit 1 buffers the far-end packet
 * (aec_setFrameFar), bit 2 processes the near-end packet (aec_process), 3 does both in the
 * reference's order (aec_process2).  d_far is the shared far-end (packet p at
 * d_far + p*far_packet_stride); near/out packet p of stream s at s*stream_stride + p*packet_stride;
 * d_out may alias d_near.  delay_ms is the reported sound-card delay (the daemon passes 0).
 * Returns 0, a WMX_E* error, or -1 where the reference wrapper would return non-zero (delay outside
 * [0,500]: the offending packet is left unwritten and nothing after it runs).
 * Float path: same operation order as the reference and the same powf (glibc's algorithm, restated
 * in csrc/libm_dev.h); tests require bit-exactness. */
typedef struct wmx_aec wmx_aec;
int wmx_aec_create(wmx_aec **out, int n_streams, int chn, int freq, int interval_ms);
int wmx_aec_destroy(wmx_aec *h);
int wmx_aec_packet_samples(const wmx_aec *h);
int wmx_aec_run(wmx_aec *h, int mode, const int16_t *d_far, long far_packet_stride, const int16_t *d_near,
                int16_t *d_out, int n_packets, long stream_stride, long packet_stride, int delay_ms, void *stream);
/* Far-end groups: the reference handle owns its far-end (aec_process2(fp, far, near, ...), src/webrtc.c:410), so a batch
 * need not share one.  n_far far-ends per batch; host array stream_far[n_streams] gives each stream's far-end in
 * [0, n_far).  All far-ends are driven in lockstep (same packet count and reported delay per call).  wmx_aec_run_groups
 * takes far-end g's packet p at d_far + g*far_group_stride + p*far_packet_stride.  wmx_aec_create / wmx_aec_run are the
 * n_far = 1 forms.  Cost: one far-end wave per group in front of the near kernel; configs with one far-end run unchanged. */
int wmx_aec_create_groups(wmx_aec **out, int n_streams, int chn, int freq, int interval_ms, int n_far, const int32_t *stream_far);
int wmx_aec_run_groups(wmx_aec *h, int mode, const int16_t *d_far, long far_packet_stride, long far_group_stride,
                       const int16_t *d_near, int16_t *d_out, int n_packets, long stream_stride, long packet_stride,
                       int delay_ms, void *stream);
int wmx_aec_state_words(const wmx_aec *h);
int wmx_aec_export_state(const wmx_aec *h, int stream_index, float *host_words);
/* Cohorts.  Everything in the AEC that decides WHERE data goes (ProcessNormal's start-up machine and delay filter, the ring
 * positions, the block counters, W:echo_cancellation.c:599-872, aec_core.c:1719-1850) depends on when the handle was
 * created and on the delays it was called with, never on the audio -- and so does the blocking of the far-end (64-sample
 * blocks counted from the handle's first packet).  Streams that were started at the same packet and report the same delay
 * form a COHORT: one control plane on the host, one far-end history on the device.  A far-end group of
 * wmx_aec_create_groups IS a cohort (stream_far may be NULL there: every stream starts in cohort 0).  A stream joins by
 *     wmx_aec_reset_cohort(h, c, stream)               -- once per cohort and join time: aec_init of the shared part
 *     wmx_aec_reset_streams(h, idx, n, c, stream)      -- aec_release + aec_init of the streams, now members of c (c = -1:
 *                                                         membership unchanged, e.g. one cohort and every stream restarted)
 * and leaves through the active mask.  wmx_aec_run_cohorts is wmx_aec_run_groups with a reported delay PER COHORT
 * (aec_process2's delayms is per handle, src/webrtc.c:410), an optional on/off byte per cohort (0: not called, control plane
 * and far history stand still) and an optional per-cohort return code (what aec_process2 would have returned to its
 * members; a rejected cohort runs nothing after the offending packet, the others carry on).  All arrays are HOST arrays of
 * wmx_aec_cohorts(h) entries.  Returns 0, WMX_E*, or the first non-zero cohort code. */
int wmx_aec_cohorts(const wmx_aec *h);
int wmx_aec_reset_cohort(wmx_aec *h, int cohort, void *stream);
/* Cohorts come and go with the handles they stand for (the reference makes a handle inside the heartbeat on first use and
 * releases it when its switch drops or recording idles, src/wmix.c:565-600, 635-636): wmx_aec_add_cohort starts a NEW one --
 * aec_init of the shared part at this point of the packet sequence; a retired id when there is one, else the next, with the
 * device buffers growing by doubling -- and returns its id in *cohort; wmx_aec_retire_cohort says that every member was
 * released: the cohort is never called again and its id may be handed out again.  wmx_aec_cohorts(h) = ids in use, retired
 * ones included = the length of the per-cohort arrays.  Per launch only packets x cohorts plans of 224 bytes cross PCIe (the
 * comfort noise's phases are not in them: they lie in a device table indexed by the stream's own block count). */
int wmx_aec_add_cohort(wmx_aec *h, int *cohort, void *stream);
int wmx_aec_retire_cohort(wmx_aec *h, int cohort);
/* Coalescing.  Handles of the reference that were created at different times but are called with the same delay end up with
 * control planes that differ only in WHERE their rings stand (W: echo_cancellation.c:599-872 and aec_core.c:1719-1850 are index
 * arithmetic on fill levels): from then on they compute the same far-end spectra, far power and plans, once per handle.  This
 * call merges such cohorts: (1) it completes the merges whose check -- launched by an earlier call -- came back equal: the far-end
 * slabs of the two cohorts were compared on the device, word for word under the rotation between their ring positions (that
 * includes the running far power, an IIR from each handle's own start); the members of `from` then get their re-blocking rings
 * rotated to the positions of `into`, its id, and `from` is retired; every stream keeps the comfort-noise generator it would have
 * as its own handle (its state is the stream's block count, part of the stream's state).  The pairs are reported in
 * merged_from[] / merged_into[] (*n_merged of them, at most cap) so that the caller can redirect its own tables; the id range
 * wmx_aec_cohorts(h) shrinks behind the last live cohort.  (2) it proposes up to max_pairs (<= 32) new pairs -- equal fill
 * levels and delay filter, start-up over, lowest id leads -- and launches their comparison behind the work already in
 * `stream`.  Nothing waits for the device: a pair proposed by one call is merged by a later one.  A pair is dropped when its two
 * cohorts are not called identically in between (delay, cohort_on, private far-end packets).  max_pairs = 0: only (1).
 * Merged streams are bound to one reported delay from then on, like the members of any cohort. */
/* cohorts of the id range that are not retired (what a launch really computes far-end spectra and plans for) */
int wmx_aec_live_cohorts(const wmx_aec *h);
int wmx_aec_coalesce(wmx_aec *h, int max_pairs, int32_t *merged_from, int32_t *merged_into, int cap, int *n_merged, void *stream);
int wmx_aec_reset_streams(wmx_aec *h, const int32_t *idx, int n, int cohort, void *stream);
int wmx_aec_set_active(wmx_aec *h, const uint8_t *host_mask, void *stream);
int wmx_aec_stream_state_bytes(const wmx_aec *h);
int wmx_aec_export_stream(wmx_aec *h, int stream_index, void *host_blob);
int wmx_aec_import_stream(wmx_aec *h, int stream_index, const void *host_blob, int cohort);
int wmx_aec_cohort_state_bytes(const wmx_aec *h);
int wmx_aec_export_cohort(wmx_aec *h, int cohort, void *host_blob);
int wmx_aec_import_cohort(wmx_aec *h, int cohort, const void *host_blob);
int wmx_aec_run_cohorts(wmx_aec *h, int mode, const int16_t *d_far, long far_packet_stride, long far_group_stride,
                        const int16_t *d_near, int16_t *d_out, int n_packets, long stream_stride, long packet_stride,
                        const int32_t *delay_ms, const uint8_t *cohort_on, int32_t *cohort_rc, void *stream);
/* In-stream timing of the two AEC kernels (bench.py's roofline entry): with timing on, every near-end launch is bracketed
 * by HIP events recorded on the launch stream -- before the far kernel, between the two, after the near kernel.
 * wmx_aec_timing waits for the last one, returns the number of launches and the summed durations (ms) since the previous
 * call and starts over. */
int wmx_aec_set_timing(wmx_aec *h, int on);
int wmx_aec_timing(wmx_aec *h, int *n_launches, double *far_ms, double *near_ms);
/* Host side of the same launches: their number and the seconds the per-cohort control planes (index arithmetic, W:
 * echo_cancellation.c:599-872 per cohort and packet) took on the caller's thread since the previous call. */
int wmx_aec_host_ctl(wmx_aec *h, long *n_launches, double *seconds);

/* ------------------------------------------------------------------ the record heartbeat: NS -> AEC -> AGC -> VAD in one call
 * wmix_shmem_write_circle (src/wmix.c:613-709) runs, per WMIX_INTERVAL_MS of captured audio and on one buffer in place,
 * ns_process -> aec_process2(far, near = out = buffer, delayms) -> agc_process -> vad_process, each behind its
 * webrtcEnable[] switch.  wmx_chain is that heartbeat for a batch: `stages` = the switches, one wmx_chain_process per tick
 * launches the enabled stages back to back on `stream` (no host synchronisation, nothing copied).
 * n10 = 10 ms packets per stream in the tick (the daemon's 20 ms tick is 2); packet p of stream s lies at
 * s*stream_stride + p*packet_stride in d_in / d_out (int16 elements; d_out == d_in is the daemon's case), the shared
 * far-end's 10 ms packet p at d_far + p*far_packet_stride.  vad_process is ONE call per tick, as in the heartbeat.  Stages
 * whose own packet is 20 ms (AEC at 8 kHz / VAD, when interval_ms % 20 == 0) need the tick contiguous (packet_stride == one
 * 10 ms packet).  delay_ms / cohort_on / cohort_rc: HOST arrays of n_cohorts entries as in wmx_aec_run_cohorts (NULL: delay 0 as
 * the daemon passes it, every cohort on, no codes wanted).  Returns 0, WMX_E*, or -1 when a cohort's delay was rejected.
 * The lifetime calls forward to every stage; the stage handles themselves are reachable for everything else. */
#define WMX_CHAIN_NS 1u
#define WMX_CHAIN_AEC 2u
#define WMX_CHAIN_AGC 4u
#define WMX_CHAIN_VAD 8u
/* The reference's two build-time alternates of the same stages (src/webrtc.c:512-521: -DMAKE_WEBRTC_NSX puts WebRtcNsx_* behind
 * ns_*; :168-191: `#undef MAKE_WEBRTC_AEC` puts WebRtcAecm_* behind aec_*), as stage bits of the one library: with WMX_CHAIN_NSX the
 * NS stage (WMX_CHAIN_NS must be set too) is the fixed-point suppressor, with WMX_CHAIN_AECM the AEC stage is the fixed-point
 * canceller.  A chain of NSX + AECM + AGC + VAD is integer end to end: bit-exact against the reference. */
#define WMX_CHAIN_NSX 16u
#define WMX_CHAIN_AECM 32u
typedef struct wmx_chain wmx_chain;
int wmx_chain_create(wmx_chain **out, int n_streams, int chn, int freq, int interval_ms, int agc_value, unsigned stages,
                     int n_cohorts);
/* stream_cohort: HOST array of n_streams entries (NULL: every stream in cohort 0) -- the cohort (control plane + far-end) each stream
 * belongs to from the start; wmx_chain_process_groups hands every cohort its OWN far-end: cohort c's 10 ms packet p at
 * d_far + c * far_group_stride + p * far_packet_stride (aec_process2's far-end is per handle, src/webrtc.c:410) */
int wmx_chain_create_groups(wmx_chain **out, int n_streams, int chn, int freq, int interval_ms, int agc_value, unsigned stages,
                            int n_cohorts, const int32_t *stream_cohort);
int wmx_chain_process_groups(wmx_chain *h, const int16_t *d_far, long far_packet_stride, long far_group_stride, const int16_t *d_in,
                             int16_t *d_out, int n10, long stream_stride, long packet_stride, const int32_t *delay_ms,
                             const uint8_t *cohort_on, int32_t *cohort_rc, void *stream);
int wmx_chain_destroy(wmx_chain *h);
/* webrtcEnable[] at run time (the daemon's message thread sets the switches, src/wmix.c:1010-1050): a stage whose bit drops is released
 * (src/wmix.c:783-813), one whose bit comes on is made anew -- fresh state for every stream, the cohorts the chain was created with,
 * agc_value as agc_init's value (< 0: the chain's own) -- stages that stay on keep their state.  stages = 0 is a heartbeat with every
 * switch off (the package passes through; such a chain works in place).  wmx_chain_create takes stages = 0 too.  A control-plane call:
 * the device is drained when a stage goes. */
int wmx_chain_set_stages(wmx_chain *h, unsigned stages, int agc_value);
unsigned wmx_chain_stages(const wmx_chain *h);
int wmx_chain_process(wmx_chain *h, const int16_t *d_far, long far_packet_stride, const int16_t *d_in, int16_t *d_out, int n10,
                      long stream_stride, long packet_stride, const int32_t *delay_ms, const uint8_t *cohort_on,
                      int32_t *cohort_rc, void *stream);
int wmx_chain_reset_streams(wmx_chain *h, const int32_t *idx, int n, int cohort, void *stream);
int wmx_chain_reset_cohort(wmx_chain *h, int cohort, void *stream);
/* wmx_chain_reset_streams with agc_init's own `value` for the new handles (the daemon creates the AGC with the volumeAgc of
 * that moment, src/wmix.c:684), and agc_addition for the listed streams of a running chain (src/wmix.c:1068-1070); a chain
 * without an AGC stage ignores the value */
int wmx_chain_reset_streams_gain(wmx_chain *h, const int32_t *idx, int n, int cohort, int agc_value, void *stream);
int wmx_chain_set_agc_gain_streams(wmx_chain *h, const int32_t *idx, int n, int agc_value, void *stream);
/* wmx_aec_add_cohort / wmx_aec_retire_cohort / wmx_aec_cohorts of the chain's AEC (a chain without one has a single cohort) */
int wmx_chain_add_cohort(wmx_chain *h, int *cohort, void *stream);
int wmx_chain_retire_cohort(wmx_chain *h, int cohort);
/* wmx_aec_coalesce / wmx_aecm_coalesce of the chain's echo canceller (a chain without one merges nothing) */
int wmx_chain_coalesce(wmx_chain *h, int max_pairs, int32_t *merged_from, int32_t *merged_into, int cap, int *n_merged, void *stream);
int wmx_chain_cohorts(const wmx_chain *h);
int wmx_chain_set_active(wmx_chain *h, const uint8_t *host_mask, void *stream);
int wmx_chain_stream_state_bytes(const wmx_chain *h);
int wmx_chain_export_stream(wmx_chain *h, int stream_index, void *host_blob);
int wmx_chain_import_stream(wmx_chain *h, int stream_index, const void *host_blob, int cohort);
wmx_ns *wmx_chain_ns(wmx_chain *h);
wmx_aec *wmx_chain_aec(wmx_chain *h);
wmx_agc *wmx_chain_agc(wmx_chain *h);
wmx_vad *wmx_chain_vad(wmx_chain *h);

/* ------------------------------------------------------------------ AECM (fixed-point echo canceller)
 * Batched form of the SAME five wrapper functions when the reference is built with its AECM switch (`#undef
 * MAKE_WEBRTC_AEC`, src/webrtc.c:168-191: WebRtcAecm_Create / Init / BufferFarend / Process, echo mode 3, comfort noise on,
 * W:modules/audio_processing/aecm/).  Same arguments, modes, strides, aliasing and return values as wmx_aec_*; one shared
 * far-end per batch.  When the reference returns -1 for a delay outside [0, 500] ms the packet HAS been processed with
 * the delay clamped (state advances) but its output is not written, as in the wrapper.  Integer path: bit-exact.  At
 * 8 kHz with 20 ms packets the reference replays an uninitialised buffer (farendOld[1]); it is zero here.
 * The legacy aec_init picks this implementation when the environment has WMIX_AMD_AECM=1. */
typedef struct wmx_aecm wmx_aecm;
int wmx_aecm_create(wmx_aecm **out, int n_streams, int chn, int freq, int interval_ms);
int wmx_aecm_destroy(wmx_aecm *h);
int wmx_aecm_packet_samples(const wmx_aecm *h);
int wmx_aecm_state_bytes(const wmx_aecm *h);
int wmx_aecm_run(wmx_aecm *h, int mode, const int16_t *d_far, long far_packet_stride, const int16_t *d_near,
                 int16_t *d_out, int n_packets, long stream_stride, long packet_stride, int delay_ms, void *stream);
/* Cohorts, as for the float AEC (see wmx_aec_run_cohorts): n_cohorts control planes + far-end histories per batch; every
 * stream starts in cohort 0.  far_group_stride: int16 elements between the far-end packets of consecutive cohorts (0: every
 * cohort hears the same far-end). */
int wmx_aecm_create_cohorts(wmx_aecm **out, int n_streams, int chn, int freq, int interval_ms, int n_cohorts);
int wmx_aecm_cohorts(const wmx_aecm *h);
int wmx_aecm_reset_cohort(wmx_aecm *h, int cohort, void *stream);
int wmx_aecm_add_cohort(wmx_aecm *h, int *cohort, void *stream); /* as wmx_aec_add_cohort / wmx_aec_retire_cohort */
int wmx_aecm_retire_cohort(wmx_aecm *h, int cohort);
/* as wmx_aec_coalesce / wmx_aec_live_cohorts: cohorts whose control planes have converged (the AECM's has no periodic counters: one
 * cohort per reported delay and phase of the 80-in-64 re-blocking remains) are merged after their far-end slabs -- far ring, frame
 * ring, the 256-block history of far spectra, Q-domains and binary spectra, the binary spectrum's running thresholds -- compared
 * equal on the device; the members' near and out frame rings are rotated; the comfort-noise generator was per stream already */
int wmx_aecm_coalesce(wmx_aecm *h, int max_pairs, int32_t *merged_from, int32_t *merged_into, int cap, int *n_merged, void *stream);
int wmx_aecm_live_cohorts(const wmx_aecm *h);
/* Why two cohorts do (not) fold: the words of the control plane that decide its future, positions taken out -- cohorts are proposed
 * for a merge when these are equal.  AEC (11): fill levels of the near / out / far-spectrum / far-sample rings, system delay, known
 * delay of the core, knownDelay, timeForDelayChange, msInSndCardBuf, filtDelay, lastDelayDiff (the core's two block counters, noise
 * floor and delay estimate, count with the stream).  AECM (8): fill levels of the far / frame / out rings, knownDelay, timeForDelayChange, msInSndCardBuf, filtDelay,
 * lastDelayDiff.  Returns 0, 1 (the cohort is retired or still in its start-up phase: no key), or WMX_E*. */
int wmx_aec_cohort_key(const wmx_aec *h, int cohort, int32_t *key11);
int wmx_aecm_cohort_key(const wmx_aecm *h, int cohort, int32_t *key8);
int wmx_aecm_reset_streams(wmx_aecm *h, const int32_t *idx, int n, int cohort, void *stream);
int wmx_aecm_set_active(wmx_aecm *h, const uint8_t *host_mask, void *stream);
int wmx_aecm_stream_state_bytes(const wmx_aecm *h);
int wmx_aecm_export_stream(wmx_aecm *h, int stream_index, void *host_blob);
int wmx_aecm_import_stream(wmx_aecm *h, int stream_index, const void *host_blob, int cohort);
int wmx_aecm_cohort_state_bytes(const wmx_aecm *h);
int wmx_aecm_export_cohort(wmx_aecm *h, int cohort, void *host_blob);
int wmx_aecm_import_cohort(wmx_aecm *h, int cohort, const void *host_blob);
int wmx_aecm_run_cohorts(wmx_aecm *h, int mode, const int16_t *d_far, long far_packet_stride, long far_group_stride,
                         const int16_t *d_near, int16_t *d_out, int n_packets, long stream_stride, long packet_stride,
                         const int32_t *delay_ms, const uint8_t *cohort_on, int32_t *cohort_rc, void *stream);
/* the fixed-point stage handles of a chain made with WMX_CHAIN_NSX / WMX_CHAIN_AECM (NULL otherwise) */
wmx_nsx *wmx_chain_nsx(wmx_chain *h);
wmx_aecm *wmx_chain_aecm(wmx_chain *h);

/* ------------------------------------------------------------------ resample + mix
 * Batched forms of wmix_pcm_zoom (src/wmix.c:139-222) and of wmix_load_data + the play thread's drain
 * (src/wmix.c:1639-1957, 1347-1366).  Strides in int16 elements.  Integer path: bit-exact, including the
 * dead 2ch->2ch branch of wmix_pcm_zoom (writes 0 bytes) and the order-dependent saturating add. */
/* out_capacity: bytes available per output row; a conversion that needs more returns WMX_EINVAL with *out_len = the
 * bytes it needs (wmix_len_of_out gives the same figure beforehand) and writes nothing. */
int wmx_pcm_zoom(int inChn, int inFreq, const int16_t *d_in, uint32_t inLen, int outChn, int outFreq, int16_t *d_out,
                 uint32_t out_capacity, long in_stride, long out_stride, int n_streams, uint32_t *out_len, void *stream);
typedef struct wmx_mix wmx_mix;
int wmx_mix_create(wmx_mix **out, int n_groups, int ring_chn, int ring_freq);
int wmx_mix_destroy(wmx_mix *m);
int wmx_mix_set(wmx_mix *m, uint32_t head_off, uint32_t tick, int reduce_mode); /* wmix->head/tick/reduceMode */
/* VIEW_PLAY_CORRECT = PLAT_PLAY_CORRECT (src/wmixPlat.h:20; src/wmix.c:1668-1669): bytes in front of the play head where a source
 * without a cursor starts.  Compile-time in the reference, per platform directory: platform/alsa/plat.h:21 chn*freq*16/8/5 (200 ms;
 * what wmx_mix_create sets), platform/hi3516/plat.h:16 and platform/t31/plat.h:16 0.  A whole number of frames inside the ring, else
 * WMX_EINVAL.  (The legacy wmix_load_data of wmix_compat.h reads WMIX_AMD_PLAY_CORRECT=<bytes> from the environment.) */
int wmx_mix_set_play_correct(wmx_mix *m, uint32_t bytes);
int wmx_mix_ring_bytes(const wmx_mix *m);
int wmx_mix_load(wmx_mix *m, const int16_t *d_src, uint32_t srcU8Len, int freq, int channels, int sample, int n_src,
                 long group_stride, long source_stride, int reduce, uint32_t *head, uint32_t *tick, void *stream);
int wmx_mix_drain(wmx_mix *m, int16_t *d_out, uint32_t bytes, long out_stride, void *stream);
int wmx_mix_export(const wmx_mix *m, int group, int16_t *host_ring, uint32_t *head_off, uint32_t *tick);

/* ------------------------------------------------------------------ math/fft.c (stand-alone radix-2 FFT helpers)
 * Batched form of FFT / FFTR / IFFT / IFFTR (math/fft.h:19-38, math/fft.c:121-398): n_batch independent
 * transforms of N points per launch, arrays laid out [n_batch][N] float on the device.  kind: 0 FFT, 1 FFTR,
 * 2 IFFT, 3 IFFTR.  As in the reference any of the arrays may be NULL: a NULL input reads as zeros, a NULL output is
 * skipped; the inverses have no amplitude / phase outputs (d_out_af / d_out_pf are ignored for kind >= 2); the real
 * variants never read d_in_im.  N must be a power of two, 2 <= N <= 4096, else WMX_EINVAL.  Bit-exact for
 * re / im / amplitude (the twiddles are the reference's double cos/sin values, tabulated on the host);
 * the phase curve is a double atan2 rounded to float. */
#define WMX_MFFT_FFT 0
#define WMX_MFFT_FFTR 1
#define WMX_MFFT_IFFT 2
#define WMX_MFFT_IFFTR 3
int wmx_mfft(int kind, int n_batch, unsigned N, const float *d_in_re, const float *d_in_im, float *d_out_re,
             float *d_out_im, float *d_out_af, float *d_out_pf, void *stream);
/* Batched fft_stream (math/fft.h:51, math/fft.c:413-424): for every stream, move pool[in_len..2*in_len) to the front,
 * append the in_len new samples behind it, transform the st_len-point pool and write the amplitude / phase curves
 * (either may be NULL).  d_in [n_streams][in_len], d_pool [n_streams][st_len] (updated in place), outputs
 * [n_streams][st_len].  Requires 2*in_len <= st_len (the reference reads past the pool otherwise). */
int wmx_mfft_stream(int n_streams, const float *d_in, unsigned in_len, float *d_pool, unsigned st_len, float *d_out_af,
                    float *d_out_pf, void *stream);

/* ------------------------------------------------------------------ RTP / G.711 packet edge
 * Egress: for n_streams senders set up like wmix_thread_rtp_send_pcma does (src/wmixTask.c:1058: v = 2, m = 1,
 * seq = timestamp = ssrc = 0; pt 8 for law WMX_LAW_A, pt 0 for WMX_LAW_U), one call produces one datagram per stream
 * from that stream's PCM: wmix_pcm_zoom(in_chn, in_freq -> out_chn, out_freq), G.711 encode, timestamp += codes /
 * out_chn, network-order header (src/rtp.c:35-70, src/rtp.h:37-75), seq += 1 -- the loop body of
 * src/wmixTask.c:1124-1143 fused into one kernel.  d_pcm rows are pcm_stride int16 apart and in_bytes long; d_packets
 * rows packet_stride bytes apart; *packet_bytes receives 12 + number of codes.
 * Ingest (stateless): payload size by payload type as rtp_recv decides it (160 for PCMA / PCMU, else 0;
 * src/rtp.c:86-95) and G711a2PCM of the payload (src/wmixTask.c:1282): d_pcm rows get 160 int16, d_pcm_bytes[s] = 320
 * or 0, d_seq_raw[s] = header bytes 2..3 as the reference leaves them (no ntohs).  Optional outputs may be NULL. */
typedef struct wmx_rtp wmx_rtp;
int wmx_rtp_create(wmx_rtp **out, int n_streams, int law);
int wmx_rtp_destroy(wmx_rtp *h);
int wmx_rtp_egress(wmx_rtp *h, int in_chn, int in_freq, const int16_t *d_pcm, uint32_t in_bytes, long pcm_stride, int out_chn,
                   int out_freq, uint8_t *d_packets, long packet_stride, uint32_t *packet_bytes, void *stream);
int wmx_rtp_ingest(int n_streams, const uint8_t *d_packets, long packet_stride, int16_t *d_pcm, long pcm_stride,
                   uint32_t *d_pcm_bytes, uint16_t *d_seq_raw, void *stream);
int wmx_rtp_export(wmx_rtp *h, int stream_index, uint16_t *seq, uint32_t *timestamp);

/* ------------------------------------------------------------------ the packet edge as a pipeline
 * What wmix_thread_rtp_recv_pcma, the record heartbeat and wmix_thread_rtp_send_pcma do for one stream per 20 ms
 * (src/wmixTask.c:1278-1316, src/wmix.c:613-709, src/wmixTask.c:1124-1143), for n_streams per step with only the 172-byte
 * RTP/PCMA datagrams (src/rtp.h:51-78) crossing PCIe: datagram -> wmx_rtp_ingest -> wmx_chain_process (two 10 ms packets, 8 kHz
 * mono) -> wmx_rtp_egress -> datagram.  A wmx_pipe owns `slots` (1 .. 16) sets of PINNED host rows for the datagrams in and out,
 * their device twins, a copy-in and a copy-out HIP stream and the events between them, all made once:
 *   wmx_pipe_in / _out / _far(h, slot)  the slot's host rows: n_streams x 172 bytes in, the same out, 160 far-end samples
 *   wmx_pipe_submit(h, d_far, &slot, stream)  queues H2D -> ingest, chain, egress on `stream` -> D2H for the NEXT slot (round robin)
 *                  and returns at once; it blocks only when that slot is still in flight from `slots` steps ago.  d_far: the
 *                  shared far-end's two 10 ms packets on the device, or NULL = the slot's own wmx_pipe_far samples
 *   wmx_pipe_wait(h, slot)  blocks until that slot's datagrams are in its out rows (slot < 0: every slot)
 *   wmx_pipe_step_resident  the three launches alone, on datagrams that are on the device already (rows in_stride / out_stride
 *                  bytes apart)
 * With three slots the H2D of step k + 1 and the D2H of step k - 1 overlap the compute of step k. */
typedef struct wmx_pipe wmx_pipe;
int wmx_pipe_create(wmx_pipe **out, int n_streams, int slots, int law, int agc_value, unsigned stages);
/* The same pipeline for a host that holds PCM: the heartbeat's own boundary is a package in host memory, worked on in place
 * (buffSrc: wmix_ai_read, src/wmix.c:609-612; ns / aec_process2 / agc / vad on it, :613-709).  A row is one package of one stream --
 * chn x freq x interval_ms, wmx_pipe_datagram_bytes() = WMIX_PKG_SIZE bytes -- the slot's far-end one package of the same format;
 * there is no ingest / egress, the chain (made with chn, freq, interval_ms, agc_value, stages) runs in place on the uploaded rows.
 * wmx_pipe_step_resident on such a pipe wants in_stride == out_stride. */
int wmx_pipe_create_pcm(wmx_pipe **out, int n_streams, int slots, int chn, int freq, int interval_ms, int agc_value, unsigned stages);
/* The same for CALLS: every stream hears a far-end of its own, as every handle of the reference does (aec_process2(fp, far, near, ..),
 * src/webrtc.c:410-483; the far-end of a call is the other party).  wmx_pipe_far(h, slot) then is n_streams rows of one package each
 * (stream-major like the near rows), and so is d_far of wmx_pipe_submit / wmx_pipe_step_resident.  Every stream is a control cohort
 * with a far-end history of its own (122 KB of device memory); the float canceller must be among the stages. */
int wmx_pipe_create_pcm_calls(wmx_pipe **out, int n_streams, int slots, int chn, int freq, int interval_ms, int agc_value, unsigned stages);
int wmx_pipe_destroy(wmx_pipe *h);
int wmx_pipe_slots(const wmx_pipe *h);
int wmx_pipe_datagram_bytes(const wmx_pipe *h);
uint8_t *wmx_pipe_in(wmx_pipe *h, int slot);
const uint8_t *wmx_pipe_out(wmx_pipe *h, int slot);
int16_t *wmx_pipe_far(wmx_pipe *h, int slot);
/* Failing clean.  A submit that fails BEFORE its first launch (taking the slot, the uploads) has advanced nothing: the same rows may be
 * submitted again.  One that fails LATER has lost the step -- the stages that ran have consumed its input, like the reference's heartbeat
 * when aec_process2 fails behind ns_process (src/wmix.c:613-709) -- and wmx_pipe_failed_steps counts it; in both cases the rotation,
 * the slots in flight and the download still owed for the previous step are as before the call, and the streams are drained.  Use ONE
 * compute stream per pipe (the download of a step is ordered behind its own egress whatever stream a later submit is given, but the
 * chain's state is not). */
int wmx_pipe_submit(wmx_pipe *h, const int16_t *d_far, int *slot, void *stream);
int wmx_pipe_wait(wmx_pipe *h, int slot);
long wmx_pipe_failed_steps(const wmx_pipe *h);
/* non-blocking wmx_pipe_wait: 1 = the rows of `slot` (< 0: of every slot) are in host memory, 0 = still on their way */
int wmx_pipe_poll(wmx_pipe *h, int slot);
int wmx_pipe_step_resident(wmx_pipe *h, const uint8_t *d_in, long in_stride, const int16_t *d_far, uint8_t *d_out, long out_stride,
                           void *stream);
wmx_chain *wmx_pipe_chain(wmx_pipe *h);
wmx_rtp *wmx_pipe_senders(wmx_pipe *h);

/* ------------------------------------------------------------------ the paced heartbeat over S streams in host memory
 * The reference's record thread is a PACED loop: one package of WMIX_INTERVAL_MS (20 ms, src/wmixConf.h:112) per tick, the tick's work
 * and a sleep adding up to WMIX_INTERVAL_MS * 1000 - 2000 us (src/wmix.c:536-538, 820: DELAY_US(intervalUs); the play thread likewise,
 * :1468-1474) -- a tick's processing must be over 2 ms before the next package is due.  wmx_rt is that tick for n_streams concurrent
 * streams whose packages lie in (pinned) host memory: the streams are cut into sub-batches of `sub_batch` streams (the last one shorter),
 * each a wmx_pipe of its own (state, chain, pinned rows, device twins) on ONE upload stream and ONE download stream, and a tick queues
 *     H2D(b) -> [ingest ->] NS -> AEC -> AGC -> VAD [-> egress] of sub-batch b on `stream` -> D2H(b)
 * for b = 0 .. B-1 back to back: the upload of sub-batch b + 1 and the download of sub-batch b - 1 run beside the compute of b, so the
 * tick's latency is one sub-batch's upload + everybody's compute + one sub-batch's download instead of the sum of all three.  The
 * download of sub-batch b is queued by the submit of b + 1, behind its noise suppressor (see wmx_pipe_submit); the last one by the wait.
 *   wmx_rt_create_pcm / _rtp   as wmx_pipe_create_pcm / wmx_pipe_create, for n_streams (long) in sub-batches; slots >= 1 sets of rows
 *   wmx_rt_batches, wmx_rt_batch_streams(b), wmx_rt_pipe(b)   the sub-batches; rows of tick slot k: wmx_pipe_in / _out(wmx_rt_pipe(h, b), k)
 *   wmx_rt_far(h, slot)        the tick's shared far-end in host memory (uploaded once per tick when d_far is NULL)
 *   wmx_rt_submit              queues one tick (the next slot, round robin) and returns; *slot = its slot
 *   wmx_rt_wait                blocks until every row of every queued tick is in host memory (wmx_rt_poll: the same without blocking)
 *   wmx_rt_tick                both: returns when the last row of the tick is in host memory -- the latency a paced host sees
 *   wmx_rt_step_resident       the tick's launches alone on rows already in HBM (row of stream s at d_rows + s * stride bytes, in place
 *                              for PCM; d_out rows for RTP)
 * Returns 0 or the first sub-batch's error; a failed sub-batch has lost its step (wmx_pipe_failed_steps), the others ran.
 * Streams whose packages are not all due at the same instant are served better as P handles of S / P streams released tick / P apart
 * (wmx_rt_submit at a group's release, wmx_rt_poll on the groups in flight): the device never idles long enough for its power
 * management to clock it down, and the latency is a third.  Measured on one MI355X, 16 kHz, 20 ms ticks, 30 000 ticks: 557 056 streams
 * in four groups, 0 misses of tick - 2 ms; 393 216 streams released all at once miss 0.04 - 0.4 % of their ticks (the host's wake-ups
 * and rare stalls of the device).
 * examples/host_paced.c is the paced loop in C (clock_nanosleep(TIMER_ABSTIME), --phases P); bench.py --paced the same from Python. */
typedef struct wmx_rt wmx_rt;
int wmx_rt_create_pcm(wmx_rt **out, long n_streams, int sub_batch, int slots, int chn, int freq, int interval_ms, int agc_value, unsigned stages);
int wmx_rt_create_rtp(wmx_rt **out, long n_streams, int sub_batch, int slots, int law, int agc_value, unsigned stages);
/* calls: every stream its own far-end (wmx_pipe_create_pcm_calls); the far rows of sub-batch b are wmx_pipe_far(wmx_rt_pipe(h, b), slot),
 * a d_far on the device holds all n_streams rows (wmx_rt_far is sub-batch 0's) */
int wmx_rt_create_pcm_calls(wmx_rt **out, long n_streams, int sub_batch, int slots, int chn, int freq, int interval_ms, int agc_value, unsigned stages);
int wmx_rt_destroy(wmx_rt *h);
/* sub-batch b on the library's own compute stream b % n (n = 1, the default: all on the caller's stream, one behind the other; n >= 2:
 * forked from the caller's stream per tick and joined to it again, so that the tail of one sub-batch's kernels overlaps the head of the
 * next -- measured: no gain, the chain's kernels are bound by vector issue) */
int wmx_rt_set_compute_streams(wmx_rt *h, int n);
int wmx_rt_batches(const wmx_rt *h);
int wmx_rt_batch_streams(const wmx_rt *h, int batch);
wmx_pipe *wmx_rt_pipe(wmx_rt *h, int batch);
int16_t *wmx_rt_far(wmx_rt *h, int slot);
int wmx_rt_submit(wmx_rt *h, const int16_t *d_far, int *slot, void *stream);
int wmx_rt_wait(wmx_rt *h);
int wmx_rt_poll(wmx_rt *h); /* non-blocking wmx_rt_wait: 1 = every row of every queued tick is in host memory, 0 = not yet */
int wmx_rt_tick(wmx_rt *h, const int16_t *d_far, int *slot, void *stream);
int wmx_rt_step_resident(wmx_rt *h, uint8_t *d_rows, long stride, const int16_t *d_far, uint8_t *d_out, long out_stride, void *stream);

/* ------------------------------------------------------------------ AEC far-end delay FIFO
 * Batched form of playPkgBuff_add/get and recordPkgBuff_add/get (src/wmix.c:432-526): n_slots packets
 * (AEC_FIFO_PKG_NUM = AEC_INTERVALMS / WMIX_INTERVAL_MS + 2, src/wmixConf.h:141) of pkg_bytes per stream, resident on
 * the device.  add: the packets of all streams (rows `stride` bytes apart) go into the slot under the cursor.
 * get: the packet the reference's index arithmetic selects for `delayms` is copied to d_out (rows `stride` bytes
 * apart) -- the arithmetic is reproduced as written, see oracle/orc_pkgfifo.c for what it amounts to.  A delay that
 * makes the reference read in front of its array returns WMX_EINVAL. */
typedef struct wmx_pkgfifo wmx_pkgfifo;
int wmx_pkgfifo_create(wmx_pkgfifo **out, int n_streams, int n_slots, int pkg_bytes, int interval_ms, int frame_bytes);
int wmx_pkgfifo_destroy(wmx_pkgfifo *h);
int wmx_pkgfifo_add(wmx_pkgfifo *h, const uint8_t *d_pkgs, long stride, void *stream);
int wmx_pkgfifo_get(wmx_pkgfifo *h, uint8_t *d_out, long stride, int delayms, void *stream);

/* ------------------------------------------------------------------ the daemon's tick: play package + record heartbeat
 * With WMIX_RECORD_PLAY_SYNC (src/wmixConf.h:144) the play thread does, per package of WMIX_INTERVAL_MS: drain the ring head
 * (src/wmix.c:1347-1366) -> playPkgBuff_add (:1419, :480-491) -> wmix_ao_write -> wmix_shmem_write_circle (:1439): ns_process ->
 * aec_process2(playPkgBuff_get(AEC_INTERVALMS), ..) -> agc_process -> vad_process on the captured package (:613-709) ->
 * wmix_pcm_zoom to 1 x 8000 (:730); the task threads feed the ring with wmix_load_data.  wmx_tick is n_groups such daemons side by
 * side: one ring, one FIFO row and one far-end per mix group, rec_per_group record streams per group (rows g * R .. g * R + R - 1)
 * whose echo canceller hears THAT group's playback, aec_delay_ms (= AEC_INTERVALMS, platform/alsa/plat.h:52: 400) late.  One mix
 * group = one control cohort with a far-end of its own.  chn / freq: the ring's and the capture's format (WMIX_CHN / WMIX_FREQ);
 * stages: WMX_CHAIN_* bits of the heartbeat.
 *   wmx_tick_load: this tick's wmix_load_data calls (= wmx_mix_load on the tick's mixer; same arguments and cursor rule).
 *   wmx_tick_play: the play side of one package.  d_play (may be NULL): n_groups rows, play_stride int16 apart <- what goes to the
 *                  sound card; the far-end package of every group (playPkgBuff_get) is left in wmx_tick_far(h): [n_groups][package]
 *                  int16 on the device.
 *   wmx_tick_record: the record side (the heartbeat).  d_rec: n_groups * rec_per_group rows of one package, rec_stride apart:
 *                  captured audio in, the chain's output out (in place, like the daemon's buffSrc).  d_rec_1x8000 (may be NULL):
 *                  rows of out_capacity bytes, out_stride int16 apart <- wmix_pcm_zoom(.., 1, 8000); *out_len = bytes per row.
 *   wmx_tick_run:  both, for audio captured beforehand (the daemon reads the capture between the two, src/wmix.c:609). */
typedef struct wmx_tick wmx_tick;
int wmx_tick_create(wmx_tick **out, int n_groups, int rec_per_group, int chn, int freq, int interval_ms, int aec_delay_ms, int agc_value,
                    unsigned stages);
int wmx_tick_destroy(wmx_tick *h);
/* the heartbeat's switches at run time (wmx_chain_set_stages on the tick's chain): the reference SHIPS with NS = 1, AGC = 1, VAD = 0,
 * AEC = 0 (src/wmix.c:1580-1584) and its message thread turns them (:1010-1050); a canceller that comes on hears one far-end per mix
 * group; 0 = a pure mix / FIFO / zoom tick */
int wmx_tick_set_stages(wmx_tick *h, unsigned stages, int agc_value);
/* webrtcEnable[WR_NS_PA] (src/wmix.c:1370-1386): on = 1 puts ns_process over the played package, in front of playPkgBuff_add (ns_init of
 * one suppressor per group now); on = 0 releases it */
int wmx_tick_play_ns(wmx_tick *h, int on);
/* wmix->rwTest (src/wmix.c:714-732), the self send-receive test: while on, wmx_tick_record loads the chain's output of each group's
 * first record stream back into the group's play ring (wmix_load_data with the heartbeat's own cursor, reduce 1) before the zoom;
 * switching it off forgets the cursor (:728-732) */
int wmx_tick_rw_test(wmx_tick *h, int on);
/* the daemon of another platform directory: PLAT_AEC_INTERVALMS (alsa 400, hi3516 700, t31 0; plat.h:14/19) is wmx_tick_create's
 * aec_delay_ms -- the FIFO gets AEC_FIFO_PKG_NUM = aec_delay_ms / interval_ms + 2 slots (src/wmixConf.h:141) -- and PLAT_PLAY_CORRECT is
 * set here (wmx_mix_set_play_correct on the tick's rings; default platform/alsa's) */
int wmx_tick_set_play_correct(wmx_tick *h, uint32_t bytes);
int wmx_tick_package_samples(const wmx_tick *h); /* int16 elements of one package of one stream = WMIX_PKG_SIZE / 2 */
int wmx_tick_load(wmx_tick *h, const int16_t *d_src, uint32_t srcU8Len, int freq, int channels, int sample, int n_src, long group_stride,
                  long source_stride, int reduce, uint32_t *head, uint32_t *tick, void *stream);
int wmx_tick_play(wmx_tick *h, int16_t *d_play, long play_stride, void *stream);
const int16_t *wmx_tick_far(const wmx_tick *h);
int wmx_tick_record(wmx_tick *h, int16_t *d_rec, long rec_stride, int16_t *d_rec_1x8000, long out_stride, uint32_t out_capacity,
                    uint32_t *out_len, void *stream);
int wmx_tick_run(wmx_tick *h, int16_t *d_play, long play_stride, int16_t *d_rec, long rec_stride, int16_t *d_rec_1x8000, long out_stride,
                 uint32_t out_capacity, uint32_t *out_len, void *stream);
wmx_mix *wmx_tick_mix(wmx_tick *h);
wmx_chain *wmx_tick_chain(wmx_tick *h);
wmx_pkgfifo *wmx_tick_fifo(wmx_tick *h);

/* Developer / test hook: the cross-lane FFT executors of the kernels, stand-alone, one transform per wavefront, in place
 * (wmix_amd/csrc/fft_debug.hip lists the kinds: Ooura rdft 128 / 256 through the LDS executor, aec_rdft_128 through the LDS,
 * 16-lane-register and one-point-per-lane executors, the SPL fixed-point real FFT of orders 7 and 8). */
int wmx_debug_fft(int kind, int n_batch, void *d_data, int32_t *d_aux, void *stream);
/* Developer / test hook: the NS kernels' table-driven log (kind 0, x >= 1) and exp (kind 1) evaluated on the host from
 * the same source (wmix_amd/csrc/libm_dev.h), for sweeping against libm without a GPU. */
int wmx_debug_ns_libm(int kind, const float *x, float *y, size_t n);
/* Developer / test hook: the NS kernels' division for ordinary operands (wmix_amd/csrc/libm_dev.h div_ordinary: the
 * compiler's own fp32 division sequence without its rescaling and special-case instructions) beside `a / b`, on the device
 * (device pointers) and compiled for the host. */
int wmx_debug_div(const float *d_a, const float *d_b, float *d_q_ordinary, float *d_q_ieee, size_t n, void *stream);
int wmx_debug_div_host(const float *a, const float *b, float *q, size_t n);
/* Same for the AEC kernel's table-driven powf: y[i] = x[i] ^ e[i]. */
int wmx_debug_pow(const float *x, const float *e, float *y, size_t n);
/* ... and evaluated on the device (device pointers; synchronises `stream`) */
int wmx_debug_pow_device(const float *d_x, const float *d_e, float *d_y, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WMIX_AMD_H */
