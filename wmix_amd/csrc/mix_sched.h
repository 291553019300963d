// mix_sched.h -- host side of the batched resample + mix: the reference's float phase recurrences (src/wmix.c:49-222,
// 1675-1939) run once per call and emitted as gather schedules for the kernels of mix.hip.  Pure C++, no HIP: mix.hip
// includes it, and so does the sanitizer driver tools_dev/san/host_ctl_san.cpp (ASan + UBSan over every format pair).
#pragma once
#include <cstdint>
#include <vector>

namespace wmx {
namespace {

// ---------------------------------------------------------------- host: the reference's phase recurrences
// src/wmix.c:49-91 / 94-136 share one loop; `want_in` selects which counter bounds it.
uint32_t len_walk(uint8_t inChn, uint16_t inFreq, uint8_t outChn, uint16_t outFreq, uint32_t limit, bool limit_is_in, bool return_in) {
    uint32_t inCount = 0, outCount = 0;
    float div, divStep = 0;
    if (inFreq < outFreq) {
        div = (float)inFreq / outFreq;
        while ((limit_is_in ? inCount : outCount) < limit) {
            outCount += outChn;
            divStep += div;
            if ((int)divStep > 0) {
                inCount += inChn;
                divStep -= 1.0;
            }
        }
    } else {
        div = (float)outFreq / inFreq;
        while ((limit_is_in ? inCount : outCount) < limit) {
            divStep += div;
            if ((int)divStep > 0) {
                outCount += outChn;
                divStep -= 1.0;
            }
            inCount += inChn;
        }
    }
    return return_in ? inCount : outCount;
}

// wmix_pcm_zoom (src/wmix.c:139-222) as a gather list: out sample i <- in sample idx[i]
void zoom_schedule(uint8_t inChn, uint16_t inFreq, uint32_t inLen, uint8_t outChn, uint16_t outFreq, std::vector<int32_t> &idx) {
    idx.clear();
    const uint8_t mode = (uint8_t)((inChn << 4) | (outChn & 0x0F));
    const int emit = (mode == 0x11 || mode == 0x21) ? 1 : (mode == 0x12 ? 2 : 0);  // 0x22 is dead code in the reference
    int32_t ip = 0;  // int16 index; the reference loops while the S16 cursor is below the byte address in + inLen
    float div, divStep = 0;
    if (inFreq < outFreq) {
        div = (float)inFreq / outFreq;
        while ((uint32_t)ip * 2u < inLen) {
            for (int e = 0; e < emit; e++) idx.push_back(ip);
            divStep += div;
            if ((int)divStep > 0) {
                ip += inChn;
                divStep -= 1.0;
            }
        }
    } else {
        div = (float)outFreq / inFreq;
        while ((uint32_t)ip * 2u < inLen) {
            divStep += div;
            if ((int)divStep > 0) {
                for (int e = 0; e < emit; e++) idx.push_back(ip);
                divStep -= 1.0;
            }
            ip += inChn;
        }
    }
}

// wmix_load_data's cursor walk (src/wmix.c:1675-1939) as a schedule: destination sample i (relative to the
// head, in ring samples) <- either source sample `src` or the k-th of n2 linear fill samples between
// source samples `src - step` and `src` (repairBuff, src/wmix.c:1854-1866).
struct LoadEntry {
    int32_t src;   // int16 index into the source buffer
    int16_t k;     // -1: plain copy, else fill sample index
    int16_t n2;    // divCount2
    int32_t step;  // int16 per source frame (for the fill: previous sample = src - step)
};

bool load_schedule(int ring_chn, int ring_freq, uint32_t srcU8Len, uint16_t freq, uint8_t channels, uint8_t sample,
                   std::vector<LoadEntry> &sch) {
    sch.clear();
    const int32_t freqErr = ring_freq - (int32_t)freq;
    uint32_t count;
    int32_t ps = 0;
    auto put = [&](int32_t s, int k, int n2, int step) { sch.push_back(LoadEntry{s, (int16_t)k, (int16_t)n2, step}); };
    if (freq == ring_freq && channels == ring_chn && sample == 16) {
        for (count = 0; count < srcU8Len;) {
            put(ps, -1, 0, 1);
            ps++;
            count += 2;
            if (ring_chn != 1) {
                put(ps, -1, 0, 1);
                ps++;
                count += 2;
            }
        }
        return true;
    }
    if (sample != 16 || (channels != 1 && channels != 2)) return true;  // the reference's empty 8/32-bit branches: nothing is written
    const int step = channels;
    float divCount = 0, divPow;
    if (freqErr < 0) {
        divPow = (float)(-freqErr) / ring_freq;
        for (count = 0; count < srcU8Len;) {
            if (divCount >= 1.0) {
                ps += step;
                divCount -= 1.0;
                count += 2 * step;
            } else {
                put(ps, -1, 0, step);
                if (ring_chn != 1) put(channels == 2 ? ps + 1 : ps, -1, 0, step);
                ps += step;
                divCount += divPow;
                count += 2 * step;
            }
        }
    } else {
        divPow = (float)freqErr / freq;
        int rc = 0, n2 = 0, rsrc = 0;
        for (count = 0; count < srcU8Len;) {
            if (divCount >= 1.0) {
                if (n2 > 64 || rc >= n2) return false;  // the reference would run off repairBuff[64]
                put(rsrc, rc, n2, step);
                if (ring_chn != 1) put(rsrc, rc, n2, step);
                divCount -= 1.0;
                rc += 1;
            } else {
                put(ps, -1, 0, step);
                if (ring_chn != 1) put(channels == 2 ? ps + 1 : ps, -1, 0, step);
                ps += step;
                divCount += divPow;
                count += 2 * step;
                if (divCount >= 1.0) {
                    n2 = (int)divCount + 1;
                    rsrc = ps;
                    rc = 0;
                }
            }
        }
    }
    return true;
}

}  // namespace
}  // namespace wmx
