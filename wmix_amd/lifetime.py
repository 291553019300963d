"""Per-stream lifetime inside a batch (include/wmix_amd.h): what the daemon does per handle -- create lazily, release when
the switch drops or recording idles, create again later (src/wmix.c:565-600, 617-618) -- for the streams of a batch."""
import numpy as np
import torch

from ._lib import check, lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Lifetime:
    """Mixed into the batch classes; `_mod` names the C module (ns, nsx, agc, vad, aec, aecm, chain)."""
    _mod = None

    def reset_streams(self, idx, cohort=None):
        """<m>_release + <m>_init for the listed streams, ordered on the current HIP stream.  AEC / AECM / chain: `cohort`
        makes them members of that control cohort (None: membership unchanged)."""
        a = np.ascontiguousarray(idx, dtype=np.int32)
        f = getattr(lib(), "wmx_%s_reset_streams" % self._mod)
        if self._mod in ("aec", "aecm", "chain"):
            rc = f(self._h, a.ctypes.data, a.size, -1 if cohort is None else int(cohort), _stream())
        else:
            assert cohort is None
            rc = f(self._h, a.ctypes.data, a.size, _stream())
        check(rc, "wmx_%s_reset_streams" % self._mod)

    def reset_cohort(self, cohort):
        check(getattr(lib(), "wmx_%s_reset_cohort" % self._mod)(self._h, int(cohort), _stream()), "reset_cohort")

    def add_cohort(self):
        """A new control cohort (AEC / chain): aec_init of the shared part now; returns its id (a retired one when there is
        one).  The per-cohort sequences of run_cohorts / process have n_cohorts entries afterwards."""
        import ctypes as C
        c = C.c_int(-1)
        check(getattr(lib(), "wmx_%s_add_cohort" % self._mod)(self._h, C.byref(c), _stream()), "add_cohort")
        n = getattr(lib(), "wmx_%s_cohorts" % self._mod)(self._h)
        if self._mod == "aec":
            self.n_far = n
        self.n_cohorts = n
        return c.value

    def retire_cohort(self, cohort):
        """Every member of the cohort was released: it is never called again, its id may be handed out again."""
        check(getattr(lib(), "wmx_%s_retire_cohort" % self._mod)(self._h, int(cohort)), "retire_cohort")

    def coalesce(self, max_pairs=32):
        """wmx_aec_coalesce / wmx_chain_coalesce: completes the merges whose device check came back equal and proposes up to
        max_pairs new ones.  Returns [(from, into), ...] of the cohorts merged by THIS call; n_cohorts is up to date afterwards."""
        import ctypes as C
        assert self._mod in ("aec", "aecm", "chain")
        fr = np.zeros(32, np.int32)
        to = np.zeros(32, np.int32)
        n = C.c_int(0)
        check(getattr(lib(), "wmx_%s_coalesce" % self._mod)(self._h, int(max_pairs), fr.ctypes.data, to.ctypes.data, 32, C.byref(n), _stream()),
              "coalesce")
        k = getattr(lib(), "wmx_%s_cohorts" % self._mod)(self._h)
        if self._mod == "aec":
            self.n_far = k
        self.n_cohorts = k
        return [(int(fr[j]), int(to[j])) for j in range(n.value)]

    def cohort_key(self, cohort):
        """The words of the cohort's control plane that decide whether it can fold into another (wmx_aec_cohort_key /
        wmx_aecm_cohort_key), or None while it has none (retired, start-up)."""
        assert self._mod in ("aec", "aecm")
        k = np.zeros(11 if self._mod == "aec" else 8, np.int32)
        rc = getattr(lib(), "wmx_%s_cohort_key" % self._mod)(self._h, int(cohort), k.ctypes.data)
        if rc == 1:
            return None
        check(rc, "cohort_key")
        return k

    def live_cohorts(self):
        """Cohorts that are not retired (wmx_aec_live_cohorts of the handle's float AEC)."""
        if self._mod == "aecm":
            return lib().wmx_aecm_live_cohorts(self._h)
        assert self._mod in ("aec", "chain")
        h = self._h if self._mod == "aec" else lib().wmx_chain_aec(self._h)
        if not h:  # a chain whose echo canceller is the fixed-point one
            return lib().wmx_aecm_live_cohorts(lib().wmx_chain_aecm(self._h))
        return lib().wmx_aec_live_cohorts(h)

    def set_active(self, mask):
        """mask: n_streams booleans (False = the stream is not called: state and PCM rows untouched) or None = all."""
        if mask is None:
            rc = getattr(lib(), "wmx_%s_set_active" % self._mod)(self._h, None, _stream())
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert m.shape == (self.n_streams,)
            rc = getattr(lib(), "wmx_%s_set_active" % self._mod)(self._h, m.ctypes.data, _stream())
        check(rc, "wmx_%s_set_active" % self._mod)

    # ---- migration between batches / GPUs: a stream's complete state as a host blob (blocking calls)
    def export_stream(self, stream_index):
        n = getattr(lib(), "wmx_%s_stream_state_bytes" % self._mod)(self._h)
        blob = np.zeros(n, np.uint8)
        check(getattr(lib(), "wmx_%s_export_stream" % self._mod)(self._h, int(stream_index), blob.ctypes.data), "export_stream")
        return blob

    def import_stream(self, stream_index, blob, cohort=None):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        f = getattr(lib(), "wmx_%s_import_stream" % self._mod)
        if self._mod in ("aec", "aecm", "chain"):
            rc = f(self._h, int(stream_index), blob.ctypes.data, -1 if cohort is None else int(cohort))
        else:
            assert cohort is None
            rc = f(self._h, int(stream_index), blob.ctypes.data)
        check(rc, "wmx_%s_import_stream" % self._mod)

    def _cohort_owner(self):
        """(module name, handle) that owns the cohorts: the AEC inside a chain, else this handle"""
        if self._mod == "chain":
            return "aec", lib().wmx_chain_aec(self._h)
        return self._mod, self._h

    def export_cohort(self, cohort):
        m, h = self._cohort_owner()
        n = getattr(lib(), "wmx_%s_cohort_state_bytes" % m)(h)
        blob = np.zeros(n, np.uint8)
        check(getattr(lib(), "wmx_%s_export_cohort" % m)(h, int(cohort), blob.ctypes.data), "export_cohort")
        return blob

    def import_cohort(self, cohort, blob):
        m, h = self._cohort_owner()
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        check(getattr(lib(), "wmx_%s_import_cohort" % m)(h, int(cohort), blob.ctypes.data), "import_cohort")
