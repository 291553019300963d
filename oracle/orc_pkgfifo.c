/* oracle/orc_pkgfifo.c -- TEST INFRASTRUCTURE ONLY (CPU checker; never on the product path).
 *
 * Restatement of the packet FIFOs that feed the AEC its delayed far-end in the daemon:
 * recordPkgBuff_add/get and playPkgBuff_add/get, src/wmix.c:432-526 (two copies of the same code).  Geometry from
 * src/wmixConf.h:112-141: a slot is one WMIX_INTERVAL_MS packet, AEC_FIFO_PKG_NUM = AEC_INTERVALMS / WMIX_INTERVAL_MS + 2
 * slots.  The index arithmetic is kept exactly as written, including what it does for a delay of AEC_INTERVALMS (the only
 * value the daemon passes, src/wmix.c:641-671): while the write cursor is below delay/interval the slot under the
 * cursor -- the oldest packet -- is returned, afterwards the slot with index delay/interval.
 * With delayms not a multiple of the interval the reference prepends the last bytes of the slot TWO before the selected
 * one (`_playPkgBuff[k - 1] - byteCount` points into slot k - 2); for k == 1 that address lies before the array
 * (undefined in the reference) and is reported as an error here.
 * Pinned against the real functions (oracle/_ref/ref_mix_driver pkgfifo) in tests/test_pkgfifo_oracle.py.
 */
#include <stdint.h>
#include <string.h>

typedef struct {
    uint8_t *slots;  /* [n_slots][pkg_bytes] */
    int n_slots, pkg_bytes, interval_ms, frame_bytes, count;
} orc_pkgfifo;

void orc_pkgfifo_init(orc_pkgfifo *f, uint8_t *storage, int n_slots, int pkg_bytes, int interval_ms, int frame_bytes)
{
    f->slots = storage;
    f->n_slots = n_slots;
    f->pkg_bytes = pkg_bytes;
    f->interval_ms = interval_ms;
    f->frame_bytes = frame_bytes;
    f->count = 0;
    memset(storage, 0, (size_t)n_slots * pkg_bytes);
}

void orc_pkgfifo_add(orc_pkgfifo *f, const uint8_t *pkg)
{
    memcpy(f->slots + (size_t)f->count++ * f->pkg_bytes, pkg, f->pkg_bytes);
    if (f->count >= f->n_slots) f->count = 0;
}

/* slot index and leading byte count of a get(delayms) at the current cursor; returns -1 for the reference's
 * out-of-array case */
int orc_pkgfifo_plan(const orc_pkgfifo *f, int delayms, int *slot, int *lead_slot, int *lead_bytes)
{
    const int frames = f->pkg_bytes / f->frame_bytes;
    int k = f->count - delayms / f->interval_ms;
    const int bytes = (int)((float)((delayms % f->interval_ms) * frames) / f->interval_ms) * f->frame_bytes;
    if (k >= f->n_slots)
        k = f->n_slots;
    else if (k < 0)
        k = 0;
    k = f->count - k;
    if (k >= f->n_slots)
        k -= f->n_slots;
    else if (k < 0)
        k += f->n_slots;
    *slot = k;
    *lead_bytes = bytes;
    *lead_slot = (k == 0 ? f->n_slots - 1 : k - 1) - 1; /* the slot whose tail `[k-1] - bytes` points into */
    if (bytes > 0 && *lead_slot < 0) return -1;
    return 0;
}

int orc_pkgfifo_get(const orc_pkgfifo *f, uint8_t *out, int delayms)
{
    int slot, lead_slot, lead;
    if (orc_pkgfifo_plan(f, delayms, &slot, &lead_slot, &lead)) return -1;
    if (lead > 0) memcpy(out, f->slots + (size_t)(lead_slot + 1) * f->pkg_bytes - lead, lead);
    memcpy(out + lead, f->slots + (size_t)slot * f->pkg_bytes, f->pkg_bytes - lead);
    return 0;
}
