// mix.hip -- wmix's resample + saturating-mix arithmetic, batched for gfx950.
//
// Replaces wmix_pcm_zoom / wmix_len_of_out / wmix_len_of_in (src/wmix.c:49-222) and the arithmetic of
// wmix_load_data + volumeAdd (src/wmix.c:1617-1957) for many independent mix groups per launch.
//
// Both reference routines drive their source/destination cursors with a float32 phase accumulator
// (`divStep += div; if ((int)divStep > 0) ... divStep -= 1.0`, `divCount += divPow; if (divCount >= 1.0)`)
// that depends only on the two rates and the length, never on the samples.  The host therefore runs the
// SAME float recurrence once per call and emits a gather schedule (one entry per destination sample);
// the kernels are pure HBM-streaming gathers: zoom = copy through the schedule, load = gather (+ the
// reference's linear "repair" fill when up-sampling) / reduce, then volumeAdd (a saturating add) into the
// group's 1 s ring.  All sources of a group are applied by the same thread in call order, so the
// order-dependent saturation (src/wmix.c:1617-1636) is reproduced without atomics.  Bit-exact.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "wmx_internal.h"
#include "../../include/wmix_compat.h"
#include "mix_sched.h"

namespace wmx {
namespace {

// ---------------------------------------------------------------- kernels
__global__ __launch_bounds__(256) void zoom_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                   const int32_t *__restrict__ idx, uint32_t n_out, long in_stride, long out_stride,
                                                   int n_streams) {
    const size_t total = (size_t)n_out * n_streams;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(t % n_out);
        const size_t s = t / n_out;
        out[s * out_stride + i] = in[s * in_stride + idx[i]];
    }
}

__device__ __forceinline__ int16_t volume_add(int16_t a, int16_t b) {  // src/wmix.c:1617-1636
    if (a == 0) return b;
    if (b == 0) return a;
    const int32_t s = (int32_t)a + b;
    return (int16_t)(s < -32768 ? -32768 : (s > 32767 ? 32767 : s));
}

// one thread = one ring sample of one group; all `n_src` sources are accumulated in order
__global__ __launch_bounds__(256) void load_kernel(int16_t *__restrict__ rings, uint32_t ring_samples, const int16_t *__restrict__ src,
                                                   const LoadEntry *__restrict__ sch, uint32_t n_out, uint32_t head_sample, int n_src,
                                                   long group_stride, long source_stride, int rdce, int n_groups) {
    const size_t total = (size_t)n_out * n_groups;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(t % n_out);
        const size_t g = t / n_out;
        const LoadEntry e = sch[i];
        uint32_t pos = head_sample + i;
        pos -= (pos >= ring_samples) ? ring_samples * (pos / ring_samples) : 0;
        int16_t *dst = rings + g * (size_t)ring_samples + pos;
        int16_t acc = *dst;
        const int16_t *sg = src + g * group_stride;
        for (int s = 0; s < n_src; s++) {
            const int16_t *p = sg + (size_t)s * source_stride;
            int16_t v;
            if (e.k < 0) {
                v = p[e.src];
            } else {
                // repairBuff[k] = prev + (k+1 times accumulated) step, float adds in the reference's order
                const int16_t prev = p[e.src - e.step];
                const float st = (float)((int)p[e.src] - (int)prev) / (float)e.n2;
                float sum = st;
                for (int j = 0; j < e.k; j++) sum += st;
                v = (int16_t)((float)prev + sum);
            }
            acc = volume_add(acc, (int16_t)(v / rdce));
        }
        *dst = acc;
    }
}

__global__ __launch_bounds__(256) void drain_kernel(int16_t *__restrict__ rings, uint32_t ring_samples, int16_t *__restrict__ out,
                                                    uint32_t n, uint32_t head_sample, long out_stride, int n_groups) {
    const size_t total = (size_t)n * n_groups;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)(t % n);
        const size_t g = t / n;
        uint32_t pos = head_sample + i;
        pos -= (pos >= ring_samples) ? ring_samples * (pos / ring_samples) : 0;
        int16_t *src = rings + g * (size_t)ring_samples + pos;
        out[g * out_stride + i] = *src;
        *src = 0;  // the play thread zeroes what it has read (src/wmix.c:1351-1352)
    }
}

// A growable device buffer owned by its (usually thread_local) object: freed when the owner dies -- a finished task
// thread of the daemon gives its staging buffers back -- except while the process is exiting (runtime_exiting()).
struct DevVec {
    void *p = nullptr;
    size_t cap = 0;
    int device = -1;
    DevVec() = default;
    DevVec(const DevVec &) = delete;
    DevVec &operator=(const DevVec &) = delete;
    int ensure(size_t bytes) {
        int dev = -1;
        WMX_HIP(hipGetDevice(&dev));
        if (dev != device && p) {  // the thread moved to another device: the old buffer is of no use there
            DeviceScope on(device);
            (void)hipFree(p);
            p = nullptr;
            cap = 0;
        }
        device = dev;
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        WMX_HIP(hipMalloc(&p, bytes));
        cap = bytes;
        return 0;
    }
    ~DevVec() {
        if (p && !runtime_exiting()) {
            DeviceScope on(device);
            (void)hipFree(p);
        }
    }
};

constexpr size_t kMappedMaxBytes = 64 * 1024;  // above this the DMA engines win: the copy path

}  // namespace

void zoom_gather_list(int inChn, int inFreq, uint32_t inLen, int outChn, int outFreq, std::vector<int32_t> &idx) {
    if (inChn == outChn && inFreq == outFreq) {
        idx.resize(inLen / 2);
        for (size_t i = 0; i < idx.size(); i++) idx[i] = (int32_t)i;
        return;
    }
    zoom_schedule((uint8_t)inChn, (uint16_t)inFreq, inLen, (uint8_t)outChn, (uint16_t)outFreq, idx);
}
}  // namespace wmx

struct wmx_mix {
    int device;  // the HIP device the state lives on (current device at create); every entry point switches to it
    int n_groups, chn, freq;
    uint32_t ring_bytes, head_off, tick, play_correct;
    uint8_t reduce_mode;
    int16_t *d_rings;
    uint8_t *h_rings = nullptr;  // set when the rings are pinned host memory mapped into the device (the legacy adapter's one ring)
    wmx::SchedCache sched;  // load schedules per source format, never rewritten (see SchedCache)
    std::vector<wmx::LoadEntry> sch;
};

extern "C" {

// ---- src/wmix.h:113-121: pure index arithmetic, identical loops to the reference
uint32_t wmix_len_of_out(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq) {
    if (inFreq == outFreq && inChn == outChn) return inLen;
    return wmx::len_walk(inChn, inFreq, outChn, outFreq, inLen, true, false);
}
uint32_t wmix_len_of_in(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t outLen) {
    if (inFreq == outFreq && inChn == outChn) return outLen;
    return wmx::len_walk(inChn, inFreq, outChn, outFreq, outLen, false, true);
}

// batched wmix_pcm_zoom: n_streams buffers of the same format and length.  Strides in int16 elements.  out_capacity =
// bytes available per output row: a conversion that needs more fails with WMX_EINVAL (and *out_len = what it needs)
// instead of overrunning the row -- the reference has no such check, its callers size `out` by wmix_len_of_out.
int wmx_pcm_zoom(int inChn, int inFreq, const int16_t *d_in, uint32_t inLen, int outChn, int outFreq, int16_t *d_out, uint32_t out_capacity,
                 long in_stride, long out_stride, int n_streams, uint32_t *out_len, void *stream) {
    using namespace wmx;
    if (!d_in || !d_out || n_streams < 1 || inChn < 1 || outChn < 1 || inFreq < 1 || outFreq < 1) {
        set_error("wmx_pcm_zoom: bad argument");
        return WMX_EINVAL;
    }
    hipStream_t s = as_stream(stream);
    if (inFreq == outFreq && inChn == outChn) {  // memcpy branch, src/wmix.c:154-158
        if (out_len) *out_len = inLen;
        if (inLen > out_capacity) {
            set_error("wmx_pcm_zoom: %u output bytes per row, capacity %u", inLen, out_capacity);
            return WMX_EINVAL;
        }
        if (inLen == 0) return 0;
        if (n_streams == 1)
            WMX_HIP(hipMemcpyAsync(d_out, d_in, inLen, hipMemcpyDeviceToDevice, s));
        else
            WMX_HIP(hipMemcpy2DAsync(d_out, out_stride * 2, d_in, in_stride * 2, inLen, n_streams, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    // the gather list depends on the format only: built and uploaded once per format and thread
    static thread_local std::vector<int32_t> idx;
    static thread_local SchedCache cache;
    const uint64_t k0 = ((uint64_t)inChn << 56) | ((uint64_t)outChn << 48) | ((uint64_t)(uint32_t)inFreq << 24) | (uint32_t)outFreq;
    const uint64_t k1 = ((uint64_t)(uint32_t)current_device() << 32) | inLen;
    SchedCache::Entry *ent = cache.find(k0, k1);
    if (!ent) {
        zoom_schedule((uint8_t)inChn, (uint16_t)inFreq, inLen, (uint8_t)outChn, (uint16_t)outFreq, idx);
        const int rc = cache.add(k0, k1, idx.data(), idx.size() * sizeof(int32_t), idx.size(), &ent);
        if (rc) return rc;
    }
    if (out_len) *out_len = (uint32_t)(ent->n * 2);
    if (ent->n * 2 > out_capacity) {
        set_error("wmx_pcm_zoom: %zu output bytes per row, capacity %u", ent->n * 2, out_capacity);
        return WMX_EINVAL;
    }
    if (ent->n == 0) return 0;
    const unsigned grid = stream_grid(ent->n * (size_t)n_streams, 256);
    hipLaunchKernelGGL(zoom_kernel, dim3(grid), dim3(256), 0, s, d_in, d_out, (const int32_t *)ent->p, (uint32_t)ent->n, in_stride,
                       out_stride, n_streams);
    WMX_LAUNCH_CHECK();
    return cache.used(ent, s);
}

// legacy host form, src/wmix.h:122-127
uint32_t wmix_pcm_zoom(uint8_t inChn, uint16_t inFreq, uint8_t *in, uint32_t inLen, uint8_t outChn, uint16_t outFreq, uint8_t *out) {
    using namespace wmx;
    static thread_local DevVec a, b;
    static thread_local MapVec ma, mb;
    if (inLen == 0 || !in || !out || !inFreq || !outFreq || !inChn || !outChn) return 0;
    const uint32_t need = wmix_len_of_out(inChn, inFreq, inLen, outChn, outFreq);  // what the reference's callers size `out` by
    uint32_t n = 0;
    // the calling thread's own non-blocking stream, and only that one is waited for (wmx_internal.h: thread_stream)
    hipStream_t ts = thread_stream();
    if ((size_t)inLen + need <= kMappedMaxBytes && ma.ensure(inLen + 16) == 0 && mb.ensure((size_t)need + 16) == 0) {
        memcpy(ma.host, in, inLen);
        if (wmx_pcm_zoom(inChn, inFreq, (const int16_t *)ma.dev, inLen, outChn, outFreq, (int16_t *)mb.dev, need + 16, 0, 0, 1, &n, ts) != 0)
            return 0;
        if (hipStreamSynchronize(ts) != hipSuccess) return 0;
        if (n) memcpy(out, mb.host, n);
        return n;
    }
    if (a.ensure(inLen + 16) || b.ensure((size_t)need + 16)) return 0;
    if (hipMemcpyAsync(a.p, in, inLen, hipMemcpyHostToDevice, ts) != hipSuccess || hipStreamSynchronize(ts) != hipSuccess) return 0;
    if (wmx_pcm_zoom(inChn, inFreq, (const int16_t *)a.p, inLen, outChn, outFreq, (int16_t *)b.p, need + 16, 0, 0, 1, &n, ts) != 0)
        return 0;
    if (n && (hipMemcpyAsync(out, b.p, n, hipMemcpyDeviceToHost, ts) != hipSuccess || hipStreamSynchronize(ts) != hipSuccess)) return 0;
    return n;
}

int wmx_mix_destroy(wmx_mix *m) {
    WMX_ON_DEVICE(m);
    if (!m) return 0;
    if (m->h_rings)
        (void)hipHostFree(m->h_rings);
    else if (m->d_rings)
        (void)hipFree(m->d_rings);
    delete m;
    return 0;
}

// n_groups rings of 1 s each in the ring format (the reference's compile-time WMIX_CHN / WMIX_FREQ)
int wmx_mix_create(wmx_mix **out, int n_groups, int ring_chn, int ring_freq) {
    using namespace wmx;
    if (!out) return WMX_EINVAL;
    *out = nullptr;
    if (n_groups < 1 || (ring_chn != 1 && ring_chn != 2) || ring_freq < 1000 || ring_freq > 96000) {
        set_error("wmx_mix_create: unsupported n_groups=%d chn=%d freq=%d", n_groups, ring_chn, ring_freq);
        return WMX_EINVAL;
    }
    wmx_mix *m = new wmx_mix();
    if ((m->device = wmx::current_device()) < 0) {
        delete m;
        return WMX_ENODEV;
    }
    m->n_groups = n_groups;
    m->chn = ring_chn;
    m->freq = ring_freq;
    m->ring_bytes = (uint32_t)(ring_chn * 2) * (uint32_t)ring_freq;  // WMIX_BUFF_SIZE, src/wmixConf.h:124
    m->head_off = 0;
    m->tick = 0;
    m->reduce_mode = 1;
    m->play_correct = (uint32_t)(ring_chn * ring_freq * 16 / 8 / 5);  // PLAT_PLAY_CORRECT, platform/alsa/plat.h:54
    m->d_rings = nullptr;
    hipError_t e = hipMalloc(&m->d_rings, (size_t)m->ring_bytes * n_groups);
    if (e == hipSuccess) e = hipMemset(m->d_rings, 0, (size_t)m->ring_bytes * n_groups);
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "hipMalloc/hipMemset(rings)", __FILE__, __LINE__);
        wmx_mix_destroy(m);
        return rc;
    }
    *out = m;
    return 0;
}

int wmx_mix_set(wmx_mix *m, uint32_t head_off, uint32_t tick, int reduce_mode) {
    WMX_ON_DEVICE(m);
    if (!m || head_off >= m->ring_bytes || reduce_mode < 1 || reduce_mode > 255) return WMX_EINVAL;
    m->head_off = head_off;
    m->tick = tick;
    m->reduce_mode = (uint8_t)reduce_mode;
    return 0;
}

// VIEW_PLAY_CORRECT (src/wmixPlat.h:20, src/wmix.c:1668-1669): how far in front of the play head a source without a cursor of its
// own starts.  A compile-time constant of the reference's platform directory: 200 ms of ring in platform/alsa (the default of
// wmx_mix_create), 0 in platform/hi3516 and platform/t31 (plat.h:16).
int wmx_mix_set_play_correct(wmx_mix *m, uint32_t bytes) {
    if (!m || bytes >= m->ring_bytes || bytes % (uint32_t)(m->chn * 2)) {
        wmx::set_error("wmx_mix_set_play_correct: %u bytes is not a whole frame inside the ring", bytes);
        return WMX_EINVAL;
    }
    m->play_correct = bytes;
    return 0;
}

int wmx_mix_ring_bytes(const wmx_mix *m) { return m ? (int)m->ring_bytes : WMX_EINVAL; }

// wmix_load_data for every group: n_src sources per group (source s of group g at d_src + g*group_stride +
// s*source_stride, int16 elements), all in the same format, all starting from the cursor (*head, *tick) like
// N task threads that begin together (head == UINT32_MAX is the reference's NULL head); they are added in
// index order.  On return *head / *tick hold the cursor every one of those sources ends with.
// NOTE: like the reference (src/wmix.c:1857,1914) the up-sampling fill reads one source frame past
// srcU8Len; the caller's buffers must make that frame readable.
int wmx_mix_load(wmx_mix *m, const int16_t *d_src, uint32_t srcU8Len, int freq, int channels, int sample, int n_src, long group_stride,
                 long source_stride, int reduce, uint32_t *head, uint32_t *tick, void *stream) {
    WMX_ON_DEVICE(m);
    using namespace wmx;
    if (!m || !head || !tick || n_src < 1) {
        set_error("wmx_mix_load: bad argument");
        return WMX_EINVAL;
    }
    if (!d_src || srcU8Len < 1) return 0;  // reference returns the head unchanged (src/wmix.c:1663-1664)
    uint32_t head_off = *head, tk = *tick;
    if (head_off == UINT32_MAX || tk < m->tick) {  // src/wmix.c:1666-1673
        head_off = m->head_off + m->play_correct;
        tk = m->tick + m->play_correct;
        if (head_off >= m->ring_bytes) head_off = 0;
    }
    const uint64_t k0 = ((uint64_t)srcU8Len << 32) | (uint32_t)freq, k1 = ((uint64_t)(uint8_t)channels << 8) | (uint8_t)sample;
    SchedCache::Entry *ent = m->sched.find(k0, k1);
    if (!ent) {
        if (!load_schedule(m->chn, m->freq, srcU8Len, (uint16_t)freq, (uint8_t)channels, (uint8_t)sample, m->sch)) {
            set_error("wmx_mix_load: rate ratio needs more than 64 fill samples (the reference overruns repairBuff here)");
            return WMX_EINVAL;
        }
        // More than one ring of output would make two threads of the launch read-modify-write the same ring sample (the
        // reference adds them one after the other); nothing in the daemon loads more than a few packets per call.
        if (m->sch.size() > m->ring_bytes / 2) {
            set_error("wmx_mix_load: %zu output samples do not fit the %u-sample ring in one call", m->sch.size(), m->ring_bytes / 2);
            return WMX_EINVAL;
        }
        const int rc = m->sched.add(k0, k1, m->sch.data(), m->sch.size() * sizeof(LoadEntry), m->sch.size(), &ent);
        if (rc) return rc;
    }
    const uint32_t n_out = (uint32_t)ent->n;
    const int rdce = (reduce == m->reduce_mode) ? 1 : m->reduce_mode;  // src/wmix.c:1675-1676
    hipStream_t s = as_stream(stream);
    if (n_out) {
        const unsigned grid = stream_grid((size_t)n_out * m->n_groups, 256);
        hipLaunchKernelGGL(load_kernel, dim3(grid), dim3(256), 0, s, m->d_rings, m->ring_bytes / 2, d_src, (const LoadEntry *)ent->p,
                           n_out, head_off / 2, n_src, group_stride, source_stride, rdce, m->n_groups);
        WMX_LAUNCH_CHECK();
        const int rcu = m->sched.used(ent, s);
        if (rcu) return rcu;
    }
    // cursor bookkeeping, src/wmix.c:1942-1956
    uint32_t tickAdd = n_out * 2, new_head = head_off + tickAdd;
    new_head %= m->ring_bytes;
    if (tk < m->tick) {
        new_head = m->head_off + tickAdd;
        tickAdd += m->tick;
        if (new_head >= m->ring_bytes) new_head -= m->ring_bytes;
    } else {
        tickAdd += tk;
    }
    *tick = tickAdd;
    *head = new_head;
    return 0;
}

// the play thread's drain (src/wmix.c:1347-1366): read `bytes` at the ring head into d_out (per group), zero what
// was read, advance head and tick.
int wmx_mix_drain(wmx_mix *m, int16_t *d_out, uint32_t bytes, long out_stride, void *stream) {
    WMX_ON_DEVICE(m);
    using namespace wmx;
    if (!m || !d_out || (bytes & 1) || bytes > m->ring_bytes) return WMX_EINVAL;
    if (bytes == 0) return 0;
    const uint32_t n = bytes / 2;
    const unsigned grid = stream_grid((size_t)n * m->n_groups, 256);
    hipLaunchKernelGGL(drain_kernel, dim3(grid), dim3(256), 0, as_stream(stream), m->d_rings, m->ring_bytes / 2, d_out, n, m->head_off / 2,
                       out_stride, m->n_groups);
    WMX_LAUNCH_CHECK();
    m->head_off = (m->head_off + bytes) % m->ring_bytes;
    m->tick += bytes;
    return 0;
}

// legacy host form, src/wmix.h:40-49.  The ring format is the reference's compile-time WMIX_CHN x WMIX_FREQ; the
// default platform is 1 x 8000 (platform/alsa/plat.h:48-50).  A differently configured daemon sets
// WMIX_AMD_RING="chn,freq" in the environment, and one built for platform/hi3516 or platform/t31 WMIX_AMD_PLAY_CORRECT=0
// (PLAT_PLAY_CORRECT in bytes, plat.h:16; unset = platform/alsa's 200 ms).
WMix_Point wmix_load_data(WMix_Struct_Head *wmix, WMix_Point src, uint32_t srcU8Len, uint16_t freq, uint8_t channels, uint8_t sample,
                          WMix_Point head, uint8_t reduce, uint32_t *tick) {
    using namespace wmx;
    WMix_Point pHead = head;
    if (!wmix || !wmix->run || !src.U8 || srcU8Len < 1) return pHead;  // src/wmix.c:1663-1664
    static int ring_chn = 0, ring_freq = 0;
    static long play_correct = -1;  // -1: the default of wmx_mix_create
    if (!ring_chn) {
        const char *pc = getenv("WMIX_AMD_PLAY_CORRECT");
        char *endp = nullptr;
        if (pc && pc[0]) {
            const long v = strtol(pc, &endp, 10);
            if (endp && !*endp && v >= 0) play_correct = v;
        }
        ring_chn = 1;
        ring_freq = 8000;
        const char *env = getenv("WMIX_AMD_RING");
        int c = 0, f = 0;
        if (env && sscanf(env, "%d,%d", &c, &f) == 2 && (c == 1 || c == 2) && f >= 1000) {
            ring_chn = c;
            ring_freq = f;
        }
    }
    const uint32_t size = (uint32_t)(wmix->end.U8 - wmix->start.U8);
    // this thread's device ring: destroyed with the thread (a finished task thread of the daemon gives it back)
    struct MixOwner {
        wmx_mix *m = nullptr;
        ~MixOwner() {
            if (m && !runtime_exiting()) wmx_mix_destroy(m);
        }
    };
    static thread_local MixOwner owner;
    static thread_local DevVec d_src;
    wmx_mix *&m = owner.m;
    if (!m || m->chn != ring_chn || m->freq != ring_freq) {
        if (m) wmx_mix_destroy(m);
        m = nullptr;
        if (wmx_mix_create(&m, 1, ring_chn, ring_freq) != 0) return pHead;
        {  // this ring only ever holds the span of one call: pinned host memory the kernel works on over PCIe (see MapVec)
            void *hp = nullptr, *dp = nullptr;
            if (hipHostMalloc(&hp, m->ring_bytes, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                (void)hipFree(m->d_rings);
                memset(hp, 0, m->ring_bytes);
                m->h_rings = static_cast<uint8_t *>(hp);
                m->d_rings = static_cast<int16_t *>(dp);
            } else {
                (void)hipGetLastError();
                if (hp) (void)hipHostFree(hp);
            }
        }
        if (play_correct >= 0 && wmx_mix_set_play_correct(m, (uint32_t)play_correct) != 0) {
            fprintf(stderr, "wmix_amd: WMIX_AMD_PLAY_CORRECT=%ld: %s\n", play_correct, wmx_last_error());
            wmx_mix_destroy(m);
            m = nullptr;
            return pHead;
        }
    }
    if (size != m->ring_bytes) {
        set_error("wmix_load_data: ring of %u bytes does not match WMIX_AMD_RING=%d,%d", size, ring_chn, ring_freq);
        fprintf(stderr, "wmix_amd: %s\n", wmx_last_error());
        return pHead;
    }
    m->head_off = (uint32_t)(wmix->head.U8 - wmix->start.U8);
    m->tick = wmix->tick;
    m->reduce_mode = wmix->reduceMode;
    uint32_t h = head.U8 ? (uint32_t)(head.U8 - wmix->start.U8) : UINT32_MAX, t = *tick;
    // The reference touches ring bytes [head, head + n_out*2) only, while other task threads and the play thread work on
    // the same ring without a lock (src/wmix.c:1347-1352, 1678-1702).  So does this adapter: the span the call will write
    // is worked out first (same cursor rule and schedule as wmx_mix_load), only that span goes up, and only it comes
    // back.  Between the two copies the adapter is a read-modify-write like the reference's per-sample `*pHead = ...`,
    // just longer: it gives no more atomicity than the reference does, and no less outside the span.
    uint32_t span_off = h;
    if (span_off == UINT32_MAX || t < m->tick) {  // src/wmix.c:1666-1673, as in wmx_mix_load
        span_off = m->head_off + m->play_correct;
        if (span_off >= m->ring_bytes) span_off = 0;
    }
    if (span_off >= size || (span_off & 1)) return pHead;
    if (!load_schedule(m->chn, m->freq, srcU8Len, freq, channels, sample, m->sch) || m->sch.size() > size / 2) {
        fprintf(stderr, "wmix_amd: wmix_load_data: unsupported rate ratio or more than one ring of output\n");
        return pHead;
    }
    const uint32_t span = (uint32_t)m->sch.size() * 2;
    const uint32_t first = span < size - span_off ? span : size - span_off, second = span - first;  // split at the wrap
    // the up-sampling fill interpolates towards the frame behind the last one (src/wmix.c:1857,1914); the copy and
    // down-sampling branches never read ahead, and neither does the adapter
    const bool reads_ahead = sample == 16 && (channels == 1 || channels == 2) && (int)freq < m->freq;
    const size_t src_bytes = (size_t)srcU8Len + (reads_ahead ? 2 * channels : 0);
    uint8_t *ring = (uint8_t *)m->d_rings;
    static thread_local MapVec m_src;
    // six task threads of the daemon load side by side (src/wmixTask.c:85, 973, 1311, 1484, 1704, 1927): each on its own non-blocking
    // stream, waiting for that stream alone (wmx_internal.h: thread_stream)
    hipStream_t ts = thread_stream();
    if (m->h_rings && src_bytes <= kMappedMaxBytes && m_src.ensure(src_bytes + 8) == 0) {
        memcpy(m_src.host, src.U8, src_bytes);
        if (first) memcpy(m->h_rings + span_off, wmix->start.U8 + span_off, first);
        if (second) memcpy(m->h_rings, wmix->start.U8, second);
        bool okm = wmx_mix_load(m, (const int16_t *)m_src.dev, srcU8Len, freq, channels, sample, 1, 0, 0, reduce, &h, &t, ts) == 0;
        okm = okm && hipStreamSynchronize(ts) == hipSuccess;
        if (!okm) {
            (void)hipGetLastError();
            fprintf(stderr, "wmix_amd: wmix_load_data failed on the GPU: %s\n", wmx_last_error());
            return pHead;
        }
        if (first) memcpy(wmix->start.U8 + span_off, m->h_rings + span_off, first);
        if (second) memcpy(wmix->start.U8, m->h_rings, second);
        *tick = t;
        pHead.U8 = wmix->start.U8 + h;
        return pHead;
    }
    bool ok = d_src.ensure(src_bytes + 8) == 0;
    ok = ok && hipMemcpyAsync(d_src.p, src.U8, src_bytes, hipMemcpyHostToDevice, ts) == hipSuccess;
    ok = ok && (!first || hipMemcpyAsync(ring + span_off, wmix->start.U8 + span_off, first, hipMemcpyHostToDevice, ts) == hipSuccess);
    ok = ok && (!second || hipMemcpyAsync(ring, wmix->start.U8, second, hipMemcpyHostToDevice, ts) == hipSuccess);
    ok = ok && wmx_mix_load(m, (const int16_t *)d_src.p, srcU8Len, freq, channels, sample, 1, 0, 0, reduce, &h, &t, ts) == 0;
    ok = ok && (!first || hipMemcpyAsync(wmix->start.U8 + span_off, ring + span_off, first, hipMemcpyDeviceToHost, ts) == hipSuccess);
    ok = ok && (!second || hipMemcpyAsync(wmix->start.U8, ring, second, hipMemcpyDeviceToHost, ts) == hipSuccess);
    ok = ok && hipStreamSynchronize(ts) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        fprintf(stderr, "wmix_amd: wmix_load_data failed on the GPU: %s\n", wmx_last_error());
        return pHead;
    }
    *tick = t;
    pHead.U8 = wmix->start.U8 + h;
    return pHead;
}

// collision-free names of the legacy group for the daemon link shim (include/wmix_compat.h, daemon_shim.c)
WMix_Point wmx_compat_load_data(WMix_Struct_Head *wmix, WMix_Point src, uint32_t srcU8Len, uint16_t freq, uint8_t channels, uint8_t sample,
                                WMix_Point head, uint8_t reduce, uint32_t *tick) {
    return wmix_load_data(wmix, src, srcU8Len, freq, channels, sample, head, reduce, tick);
}
uint32_t wmx_compat_len_of_out(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq) {
    return wmix_len_of_out(inChn, inFreq, inLen, outChn, outFreq);
}
uint32_t wmx_compat_len_of_in(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t outLen) {
    return wmix_len_of_in(inChn, inFreq, outChn, outFreq, outLen);
}
uint32_t wmx_compat_pcm_zoom(uint8_t inChn, uint16_t inFreq, uint8_t *in, uint32_t inLen, uint8_t outChn, uint16_t outFreq, uint8_t *out) {
    return wmix_pcm_zoom(inChn, inFreq, in, inLen, outChn, outFreq, out);
}

int wmx_mix_export(const wmx_mix *m, int group, int16_t *host_ring, uint32_t *head_off, uint32_t *tick) {
    WMX_ON_DEVICE(m);
    if (!m || group < 0 || group >= m->n_groups) return WMX_EINVAL;
    if (host_ring) {
        WMX_HIP(hipDeviceSynchronize());
        WMX_HIP(hipMemcpy(host_ring, (const uint8_t *)m->d_rings + (size_t)group * m->ring_bytes, m->ring_bytes, hipMemcpyDeviceToHost));
    }
    if (head_off) *head_off = m->head_off;
    if (tick) *tick = m->tick;
    return 0;
}

}  // extern "C"
