// compat_g711.hip -- the reference's G.711 entry points (src/g711codec.h:24-34,
// src/g711codec.c:194-308) exported unchanged over HOST buffers.  Each call stages
// the buffer in HBM, runs the batched kernel of g711.hip and copies the result
// back; there is no CPU arithmetic here.  Error behaviour follows the reference:
// the PCM2G711x/G711x2PCM null check only fires when in, out AND len are all
// null/0 (src/g711codec.c:230 uses &&); otherwise the element count (encode) or
// byte count (decode) is returned.  A HIP failure returns -1 and sets
// wmx_last_error().
#include <cstring>
#include "wmx_internal.h"
#include "../../include/wmix_compat.h"

namespace {

struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n < 4096 ? 4096 : n;
        WMX_HIP(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
};
thread_local Scratch g_a, g_b;

// The daemon converts one RTP payload per call (160 - 320 samples, src/wmixTask.c:285, 1139, 1282): two runtime copies around the
// launch cost several times the work.  Up to kMappedMax elements the caller's data goes through a pinned host buffer that is mapped
// into the device: memcpy in, ONE launch whose kernel reads and writes it over PCIe, one synchronisation, memcpy out.
constexpr size_t kMappedMax = 16384;
struct Mapped {
    uint8_t *host = nullptr, *dev = nullptr;  // 3 bytes per element: the int16 side at [0, 2 * kMappedMax), the codes behind it
    int ensure() {
        if (host) return 0;
        void *hp = nullptr, *dp = nullptr;
        WMX_HIP(hipHostMalloc(&hp, 3 * kMappedMax, hipHostMallocMapped | hipHostMallocPortable));
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(hp);
            return -1;
        }
        host = static_cast<uint8_t *>(hp);
        dev = static_cast<uint8_t *>(dp);
        return 0;
    }
    ~Mapped() {
        if (host && !wmx::runtime_exiting()) (void)hipHostFree(host);
    }
};
thread_local Mapped g_m;

// every launch and copy of a call goes to the calling THREAD's own non-blocking stream and only that stream is waited for
// (wmx_internal.h: thread_stream) -- the daemon's RTP threads convert side by side (src/wmixTask.c:285, 1139, 1282)
bool copy_sync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s, bool wait) {
    if (hipMemcpyAsync(dst, src, bytes, kind, s) != hipSuccess) return false;
    return !wait || hipStreamSynchronize(s) == hipSuccess;
}

int host_encode(int law, unsigned char *out, const short *in, int len) {
    if (len <= 0) return 0;
    size_t n = (size_t)len;
    hipStream_t s = wmx::thread_stream();
    if (n <= kMappedMax && g_m.ensure() == 0) {
        memcpy(g_m.host, in, n * 2);
        if (wmx_g711_encode(law, (const int16_t *)g_m.dev, g_m.dev + 2 * kMappedMax, n, s)) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        memcpy(out, g_m.host + 2 * kMappedMax, n);
        return len;
    }
    if (g_a.ensure(n * 2) || g_b.ensure(n)) return -1;
    if (!copy_sync(g_a.p, in, n * 2, hipMemcpyHostToDevice, s, true)) return -1;  // (pageable source: waited for before the caller may reuse it)
    if (wmx_g711_encode(law, (const int16_t *)g_a.p, (uint8_t *)g_b.p, n, s)) return -1;
    if (!copy_sync(out, g_b.p, n, hipMemcpyDeviceToHost, s, true)) return -1;
    return len;
}

int host_decode(int law, short *out, const unsigned char *in, int bytes) {
    if (bytes <= 0) return 0;
    size_t n = (size_t)bytes;
    hipStream_t s = wmx::thread_stream();
    if (n <= kMappedMax && g_m.ensure() == 0) {
        memcpy(g_m.host + 2 * kMappedMax, in, n);
        if (wmx_g711_decode(law, g_m.dev + 2 * kMappedMax, (int16_t *)g_m.dev, n, s)) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        memcpy(out, g_m.host, n * 2);
        return bytes * 2;
    }
    if (g_a.ensure(n) || g_b.ensure(n * 2)) return -1;
    if (!copy_sync(g_a.p, in, n, hipMemcpyHostToDevice, s, true)) return -1;
    if (wmx_g711_decode(law, (const uint8_t *)g_a.p, (int16_t *)g_b.p, n, s)) return -1;
    if (!copy_sync(out, g_b.p, n * 2, hipMemcpyDeviceToHost, s, true)) return -1;
    return bytes * 2;
}

bool all_null(const void *a, const void *b, int n) { return !a && !b && n == 0; }

}  // namespace

extern "C" {

int g711a_encode(unsigned char g711_data[], const short amp[], int len) { return host_encode(WMX_LAW_A, g711_data, amp, len); }
int g711u_encode(unsigned char g711_data[], const short amp[], int len) { return host_encode(WMX_LAW_U, g711_data, amp, len); }
int g711a_decode(short amp[], const unsigned char d[], int bytes) { return host_decode(WMX_LAW_A, amp, d, bytes); }
int g711u_decode(short amp[], const unsigned char d[], int bytes) { return host_decode(WMX_LAW_U, amp, d, bytes); }

int PCM2G711a(char *in, char *out, int DataLen, int reserve) {
    (void)reserve;
    if (all_null(in, out, DataLen)) {
        printf("Error, empty data or transmit failed, exit !\n");
        return -1;
    }
    return host_encode(WMX_LAW_A, (unsigned char *)out, (const short *)in, DataLen / 2);
}
int PCM2G711u(char *in, char *out, int DataLen, int reserve) {
    (void)reserve;
    if (all_null(in, out, DataLen)) {
        printf("Error, empty data or transmit failed, exit !\n");
        return -1;
    }
    return host_encode(WMX_LAW_U, (unsigned char *)out, (const short *)in, DataLen / 2);
}
int G711a2PCM(char *in, char *out, int DataLen, int reserve) {
    (void)reserve;
    if (all_null(in, out, DataLen)) {
        printf("Error, empty data or transmit failed, exit !\n");
        return -1;
    }
    return host_decode(WMX_LAW_A, (short *)out, (const unsigned char *)in, DataLen);
}
int G711u2PCM(char *in, char *out, int DataLen, int reserve) {
    (void)reserve;
    if (all_null(in, out, DataLen)) {
        printf("Error, empty data or transmit failed, exit !\n");
        return -1;
    }
    return host_decode(WMX_LAW_U, (short *)out, (const unsigned char *)in, DataLen);
}

// src/g711codec.c:82,120 export these although the header does not declare them.
unsigned char linear2alaw(int pcm_val) {
    short s = (short)pcm_val;
    unsigned char c = 0;
    host_encode(WMX_LAW_A, &c, &s, 1);
    return c;
}
unsigned char linear2ulaw(int pcm_val) {
    short s = (short)pcm_val;
    unsigned char c = 0;
    host_encode(WMX_LAW_U, &c, &s, 1);
    return c;
}

}  // extern "C"
