// rtp.hip -- the RTP / G.711 packet edge of the hot path, batched for gfx950.
//
// Egress replaces, for n streams per launch, the loop body of wmix_thread_rtp_send_pcma (src/wmixTask.c:1124-1143):
//   wmix_pcm_zoom (src/wmix.c:139-222) -> PCM2G711a (src/g711codec.c:227-247) -> header.timestamp += codes / chn ->
//   rtp_send's network-order header (src/rtp.c:35-70, layout src/rtp.h:37-75) -> header.seq++
// as ONE kernel: the PCM a stream produced is read once and what leaves is the datagram (12 + 160 bytes for the
// reference's 20 ms of 8 kHz A-law), so 172 B per stream per packet cross PCIe instead of 320 B of PCM plus a host
// encode.  Ingest replaces rtp_recv's payload-size rule + G711a2PCM (src/rtp.c:86-95, src/wmixTask.c:1278-1282).
// The zoom's float32 phase walk is data independent and runs once per call on the host (mix.hip) -- the kernel
// gathers through the list.  Integer path: bit-exact.
#include <vector>
#include "wmx_internal.h"
#include "g711_dev.h"

struct wmx_rtp {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_streams, law;
    uint32_t *d_seq;  // per stream: sequence number (low 16 bits significant)
    uint32_t *d_ts;   // per stream: timestamp
    wmx::SchedCache sched;  // gather list per egress format, never rewritten (see SchedCache)
    std::vector<int32_t> idx;
};

namespace wmx {
namespace {

constexpr int kRtpHeader = 12;      // RTP_HEADER_SIZE, src/rtp.h:33
constexpr int kRtpG711Payload = 160;  // RTP_PCMA_PKT_SIZE, src/rtp.h:31

template <int LAW>
__global__ void rtp_egress_kernel(const int16_t *__restrict__ pcm, long pcm_stride, const int32_t *__restrict__ idx, int n_codes,
                                  int codes_per_ts, uint32_t *seq, uint32_t *ts, uint8_t *packets, long packet_stride, int n_streams,
                                  int pt) {
    const int stream = blockIdx.y;
    if (stream >= n_streams) return;
    const int16_t *src = pcm + (size_t)stream * pcm_stride;
    uint8_t *pkt = packets + (size_t)stream * packet_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_codes; i += gridDim.x * blockDim.x)
        pkt[kRtpHeader + i] = (uint8_t)(LAW == WMX_LAW_A ? enc_alaw(src[idx[i]]) : enc_ulaw(src[idx[i]]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t t = ts[stream] + (uint32_t)codes_per_ts;  // timestamp += ret / chn, before the send
        const uint32_t s = seq[stream] & 0xFFFFu;
        pkt[0] = 2u << 6;                 // v = 2, p = x = 0, cc = 0
        pkt[1] = (uint8_t)(0x80u | pt);   // m = 1
        pkt[2] = (uint8_t)(s >> 8);       // htons / htonl: big endian on the wire
        pkt[3] = (uint8_t)s;
        pkt[4] = (uint8_t)(t >> 24);
        pkt[5] = (uint8_t)(t >> 16);
        pkt[6] = (uint8_t)(t >> 8);
        pkt[7] = (uint8_t)t;
        pkt[8] = pkt[9] = pkt[10] = pkt[11] = 0;  // ssrc = 0 (src/wmixTask.c:1058)
        ts[stream] = t;
        seq[stream] = (s + 1) & 0xFFFFu;  // rtpHeader.seq++ after the send (uint16 wrap)
    }
}

__global__ void rtp_ingest_kernel(const uint8_t *__restrict__ packets, long packet_stride, int16_t *pcm, long pcm_stride, uint32_t *pcm_bytes,
                                  uint16_t *seq_raw, int n_streams) {
    const int stream = blockIdx.x;
    if (stream >= n_streams) return;
    const uint8_t *pkt = packets + (size_t)stream * packet_stride;
    const int pt = pkt[1] & 0x7F;
    const int size = (pt == 8 || pt == 0) ? kRtpG711Payload : 0;  // src/rtp.c:88-95 (AAC-tagged packets: not G.711, size 0 here)
    int16_t *dst = pcm + (size_t)stream * pcm_stride;
    for (int i = threadIdx.x; i < size; i += blockDim.x) dst[i] = (int16_t)dec_alaw(pkt[kRtpHeader + i]);  // G711a2PCM whatever the pt
    if (threadIdx.x == 0) {
        if (pcm_bytes) pcm_bytes[stream] = (uint32_t)size * 2;
        if (seq_raw) seq_raw[stream] = (uint16_t)(pkt[2] | (pkt[3] << 8));  // as stored: rtp_recv does not ntohs
    }
}

// ---- the same two kernels for the layouts the hosts use: packets and PCM rows on 4 / 8-byte boundaries.  One lane takes FOUR
// codes (one 32-bit word of payload, one 8-byte store of samples; or four gathered samples, one 32-bit store of codes), and
// the lanes are dealt over (stream, word) pairs instead of one workgroup per stream: 65 536 streams were 65 536 workgroups of
// 64 / 256 lanes moving one byte each (115 us for the ingest, 40 for the egress); now 1.3 M lanes of four codes.
constexpr int kRtpWords = kRtpG711Payload / 4;
__global__ __launch_bounds__(256) void rtp_ingest_wide_kernel(const uint8_t *__restrict__ packets, long packet_stride, int16_t *pcm,
                                                               long pcm_stride, uint32_t *pcm_bytes, uint16_t *seq_raw, int n_streams) {
    const size_t total = (size_t)n_streams * kRtpWords;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int stream = (int)(t / kRtpWords), c = (int)(t - (size_t)stream * kRtpWords);
        const uint8_t *pkt = packets + (size_t)stream * packet_stride;
        const uint32_t h0 = *reinterpret_cast<const uint32_t *>(pkt);  // bytes 0..3 of the header: flags, pt, seq as stored
        const int pt = (h0 >> 8) & 0x7F;
        const bool g711 = pt == 8 || pt == 0;  // src/rtp.c:88-95 (AAC-tagged packets: not G.711, size 0 here)
        if (g711) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(pkt + kRtpHeader + 4 * c);
            uint2 o;  // G711a2PCM whatever the pt
            o.x = ((uint32_t)dec_alaw(w & 0xFF) & 0xFFFFu) | ((uint32_t)dec_alaw((w >> 8) & 0xFF) << 16);
            o.y = ((uint32_t)dec_alaw((w >> 16) & 0xFF) & 0xFFFFu) | ((uint32_t)dec_alaw(w >> 24) << 16);
            *reinterpret_cast<uint2 *>(pcm + (size_t)stream * pcm_stride + 4 * c) = o;
        }
        if (c == 0) {
            if (pcm_bytes) pcm_bytes[stream] = g711 ? (uint32_t)kRtpG711Payload * 2 : 0u;
            if (seq_raw) seq_raw[stream] = (uint16_t)(h0 >> 16);  // pkt[2] | pkt[3] << 8, as stored: rtp_recv does not ntohs
        }
    }
}

template <int LAW>
__global__ __launch_bounds__(256) void rtp_egress_wide_kernel(const int16_t *__restrict__ pcm, long pcm_stride, const int32_t *__restrict__ idx,
                                                               int n_codes, int codes_per_ts, uint32_t *seq, uint32_t *ts, uint8_t *packets,
                                                               long packet_stride, int n_streams, int pt) {
    const int words = n_codes >> 2;  // n_codes is a multiple of 4 here
    const size_t total = (size_t)n_streams * words;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int stream = (int)(t / words), j = (int)(t - (size_t)stream * words);
        const int16_t *src = pcm + (size_t)stream * pcm_stride;
        uint8_t *pkt = packets + (size_t)stream * packet_stride;
        const int4 ix = *reinterpret_cast<const int4 *>(idx + 4 * j);
        auto enc = [](int v) -> uint32_t { return LAW == WMX_LAW_A ? enc_alaw(v) : enc_ulaw(v); };
        *reinterpret_cast<uint32_t *>(pkt + kRtpHeader + 4 * j) =
            (enc(src[ix.x]) & 0xFF) | ((enc(src[ix.y]) & 0xFF) << 8) | ((enc(src[ix.z]) & 0xFF) << 16) | (enc(src[ix.w]) << 24);
        if (j == 0) {
            const uint32_t tt = ts[stream] + (uint32_t)codes_per_ts;  // timestamp += ret / chn, before the send
            const uint32_t sq = seq[stream] & 0xFFFFu;
            uint32_t *hd = reinterpret_cast<uint32_t *>(pkt);
            // v = 2, p = x = 0, cc = 0 | m = 1, pt | seq, timestamp big endian on the wire | ssrc = 0 (src/wmixTask.c:1058)
            hd[0] = (2u << 6) | ((0x80u | (uint32_t)pt) << 8) | ((sq >> 8) << 16) | ((sq & 0xFFu) << 24);
            hd[1] = (tt >> 24) | (((tt >> 16) & 0xFFu) << 8) | (((tt >> 8) & 0xFFu) << 16) | ((tt & 0xFFu) << 24);
            hd[2] = 0;
            ts[stream] = tt;
            seq[stream] = (sq + 1) & 0xFFFFu;  // rtpHeader.seq++ after the send (uint16 wrap)
        }
    }
}

inline bool aligned_to(const void *p, long stride_bytes, int a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0 && stride_bytes % a == 0; }

}  // namespace
}  // namespace wmx

using namespace wmx;

extern "C" {

int wmx_rtp_create(wmx_rtp **out, int n_streams, int law) {
    if (!out || n_streams < 1 || (law != WMX_LAW_A && law != WMX_LAW_U)) {
        set_error("wmx_rtp_create: n_streams %d, law %d", n_streams, law);
        return WMX_EINVAL;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
        set_error("wmx_rtp_create: no HIP device");
        return WMX_ENODEV;
    }
    wmx_rtp *h = new wmx_rtp();
    if ((h->device = wmx::current_device()) < 0) {
        delete h;
        return WMX_ENODEV;
    }
    h->n_streams = n_streams;
    h->law = law;
    h->d_seq = h->d_ts = nullptr;
    hipError_t e = hipMalloc(&h->d_seq, sizeof(uint32_t) * n_streams);
    if (e == hipSuccess) e = hipMalloc(&h->d_ts, sizeof(uint32_t) * n_streams);
    if (e == hipSuccess) e = hipMemset(h->d_seq, 0, sizeof(uint32_t) * n_streams);  // rtp_header(..., seq 0, timestamp 0, ssrc 0)
    if (e == hipSuccess) e = hipMemset(h->d_ts, 0, sizeof(uint32_t) * n_streams);
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "wmx_rtp_create: hipMalloc/hipMemset", __FILE__, __LINE__);
        wmx_rtp_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

int wmx_rtp_destroy(wmx_rtp *h) {
    WMX_ON_DEVICE(h);
    if (!h) return 0;
    if (h->d_seq) (void)hipFree(h->d_seq);
    if (h->d_ts) (void)hipFree(h->d_ts);
    delete h;
    return 0;
}

int wmx_rtp_egress(wmx_rtp *h, int in_chn, int in_freq, const int16_t *d_pcm, uint32_t in_bytes, long pcm_stride, int out_chn,
                   int out_freq, uint8_t *d_packets, long packet_stride, uint32_t *packet_bytes, void *stream) {
    WMX_ON_DEVICE(h);
    if (!h || !d_pcm || !d_packets || in_chn < 1 || in_chn > 2 || out_chn < 1 || out_chn > 2 || in_freq < 1 || out_freq < 1 ||
        in_bytes == 0 || (h->n_streams > 1 && pcm_stride < (long)(in_bytes / 2))) {
        set_error("wmx_rtp_egress: bad arguments");
        return WMX_EINVAL;
    }
    const uint64_t k0 = ((uint64_t)in_chn << 56) | ((uint64_t)out_chn << 48) | ((uint64_t)(uint32_t)in_freq << 24) | (uint32_t)out_freq;
    SchedCache::Entry *ent = h->sched.find(k0, in_bytes);
    if (!ent) {
        zoom_gather_list(in_chn, in_freq, in_bytes, out_chn, out_freq, h->idx);
        const int rc = h->sched.add(k0, in_bytes, h->idx.data(), h->idx.size() * sizeof(int32_t), h->idx.size(), &ent);
        if (rc) return rc;
    }
    const int32_t *d_idx = (const int32_t *)ent->p;
    const int n_codes = (int)ent->n;  // PCM2G711x returns DataLen / 2 codes
    if (packet_stride < kRtpHeader + n_codes) {
        set_error("wmx_rtp_egress: packet_stride %ld < %d", packet_stride, kRtpHeader + n_codes);
        return WMX_EINVAL;
    }
    if (packet_bytes) *packet_bytes = (uint32_t)(kRtpHeader + n_codes);
    const dim3 block(256), grid((unsigned)((n_codes + 255) / 256 > 0 ? (n_codes + 255) / 256 : 1), (unsigned)h->n_streams);
    const int pt = h->law == WMX_LAW_A ? 8 : 0;  // RTP_PAYLOAD_TYPE_PCMA / PCMU, src/rtp.h:21-24
    if (n_codes >= 4 && n_codes % 4 == 0 && aligned_to(d_packets, packet_stride, 4)) {  // four codes per lane (the gather list is 16-byte aligned)
        const unsigned wgrid = wmx::stream_grid((size_t)h->n_streams * (n_codes / 4), 256);
        if (h->law == WMX_LAW_A)
            hipLaunchKernelGGL((rtp_egress_wide_kernel<WMX_LAW_A>), dim3(wgrid), block, 0, as_stream(stream), d_pcm, pcm_stride, d_idx, n_codes,
                               n_codes / out_chn, h->d_seq, h->d_ts, d_packets, packet_stride, h->n_streams, pt);
        else
            hipLaunchKernelGGL((rtp_egress_wide_kernel<WMX_LAW_U>), dim3(wgrid), block, 0, as_stream(stream), d_pcm, pcm_stride, d_idx, n_codes,
                               n_codes / out_chn, h->d_seq, h->d_ts, d_packets, packet_stride, h->n_streams, pt);
    } else if (h->law == WMX_LAW_A)
        hipLaunchKernelGGL((rtp_egress_kernel<WMX_LAW_A>), grid, block, 0, as_stream(stream), d_pcm, pcm_stride, d_idx, n_codes,
                           n_codes / out_chn, h->d_seq, h->d_ts, d_packets, packet_stride, h->n_streams, pt);
    else
        hipLaunchKernelGGL((rtp_egress_kernel<WMX_LAW_U>), grid, block, 0, as_stream(stream), d_pcm, pcm_stride, d_idx, n_codes,
                           n_codes / out_chn, h->d_seq, h->d_ts, d_packets, packet_stride, h->n_streams, pt);
    WMX_LAUNCH_CHECK();
    return h->sched.used(ent, as_stream(stream));
}

int wmx_rtp_ingest(int n_streams, const uint8_t *d_packets, long packet_stride, int16_t *d_pcm, long pcm_stride, uint32_t *d_pcm_bytes,
                   uint16_t *d_seq_raw, void *stream) {
    if (n_streams < 0 || !d_packets || !d_pcm || packet_stride < kRtpHeader + kRtpG711Payload || (n_streams > 1 && pcm_stride < kRtpG711Payload)) {
        set_error("wmx_rtp_ingest: bad arguments");
        return WMX_EINVAL;
    }
    if (n_streams == 0) return 0;
    if (aligned_to(d_packets, packet_stride, 4) && aligned_to(d_pcm, pcm_stride * 2, 8))
        hipLaunchKernelGGL(rtp_ingest_wide_kernel, dim3(wmx::stream_grid((size_t)n_streams * kRtpWords, 256)), dim3(256), 0, as_stream(stream),
                           d_packets, packet_stride, d_pcm, pcm_stride, d_pcm_bytes, d_seq_raw, n_streams);
    else
        hipLaunchKernelGGL(rtp_ingest_kernel, dim3((unsigned)n_streams), dim3(64), 0, as_stream(stream), d_packets, packet_stride, d_pcm,
                           pcm_stride, d_pcm_bytes, d_seq_raw, n_streams);
    WMX_LAUNCH_CHECK();
    return 0;
}

int wmx_rtp_export(wmx_rtp *h, int stream_index, uint16_t *seq, uint32_t *timestamp) {
    WMX_ON_DEVICE(h);
    if (!h || stream_index < 0 || stream_index >= h->n_streams) return WMX_EINVAL;
    uint32_t s = 0, t = 0;
    WMX_HIP(hipDeviceSynchronize());
    WMX_HIP(hipMemcpy(&s, h->d_seq + stream_index, 4, hipMemcpyDeviceToHost));
    WMX_HIP(hipMemcpy(&t, h->d_ts + stream_index, 4, hipMemcpyDeviceToHost));
    if (seq) *seq = (uint16_t)s;
    if (timestamp) *timestamp = t;
    return 0;
}

}  // extern "C"
