"""Random restarts of ``vireo_wrap`` on one GPU (and its share of them under a communicator).

The reference builds one ``Vireo`` object per restart -- each constructor draws
``rand(n_cell, n_donor)`` and ``rand(n_var, n_donor, n_GT)`` from NumPy's global legacy
stream (vireoSNP/utils/vireo_wrap.py:66-71, vireo_model.py:98,103) -- fits them one after the
other (or in a multiprocessing.Pool) and keeps ``argmax(ELBO_[-1])`` (vireo_wrap.py:84-91).

Here a rank owns restarts ``rank, rank + world, ...``.  It walks the WHOLE stream so that every
restart sees exactly the reference's draws, but forms doubles only for its own restarts (the C
continuation of the Mersenne Twister skips the others at a fraction of a nanosecond per
double), uploads the raw draws, normalises them on the device in NumPy's summation order, and
runs all its restarts through ONE device model whose best state so far stays in HBM.
"""
import ctypes as C
import threading
import time

import numpy as np

import os

from . import _lib
from .engine import DeviceBatch, DeviceModel


# Optional phase timers (bench.py's c4 leg): set to a dict and the restart loop adds the
# seconds it spends drawing / skipping random numbers, uploading + normalising, fitting, and
# refining the winner.  None = no timing, no overhead.
PHASES = None


class _phase:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t0 = time.perf_counter() if PHASES is not None else 0.0

    def __exit__(self, *exc):
        if PHASES is not None:
            PHASES[self.name] = PHASES.get(self.name, 0.0) + time.perf_counter() - self.t0


class LegacyStream:
    """``np.random``'s global RandomState, continued by libvireo_hip.so
    (vrx_mt19937_random_sample).  Reads and writes the global state on every call, so it can be
    mixed freely with ``np.random.*`` calls."""

    @staticmethod
    def _advance(out, n):
        kind, key, pos, has_gauss, gauss = np.random.get_state()
        if kind != "MT19937":
            raise _lib.VrxError("np.random is not the legacy MT19937 stream")
        key = np.ascontiguousarray(key, dtype=np.uint32).copy()
        cpos = C.c_int32(int(pos))
        _lib.check(_lib.lib().vrx_mt19937_random_sample(
            key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(cpos), _lib.dptr(out), int(n)))
        np.random.set_state((kind, key, cpos.value, has_gauss, gauss))

    def rand(self, *shape):
        """== np.random.rand(*shape), bit for bit"""
        out = np.empty(shape)
        with _phase("draw"):
            self._advance(out.reshape(-1), out.size)
        return out

    def skip(self, n):
        """consume n doubles without forming them"""
        with _phase("skip"):
            self._advance(None, n)


def restart_batch(n_donor, n_owned, nnz, wide=True):
    """How many restarts share one device model (vrx_model_cfg.n_batch).

    A sweep of the entry stream computes 16 columns whatever n_donor is; restarts packed side by
    side fill whole sweeps.  Cost of a restart-iteration in sweeps, from the measurements in
    DESIGN_HISTORY.md section 4.4: ceil(R * n_donor / 16) sweeps -- 1.25x each when the column count is odd
    (element-wise staging of the dense operand) -- divided by R, plus 10 % for a batch running
    until its slowest restart has stopped.  The smallest R within 5 % of the best wins
    (n_donor = 16: 1; 12: 4; 8: 2; 5: 6; 4: 4).  ``wide`` = False (pair-word streams, whose
    column-block kernels are not tuned for strided operands): at most one sweep.  A problem small
    enough to be bound by kernel launches rather than by the stream takes a full batch of 16.
    VIREO_RESTART_BATCH overrides (1 = one restart at a time)."""
    n_owned = max(1, int(n_owned))  # (a rank of a shard wider than n_init owns none)
    forced = int(os.environ.get("VIREO_RESTART_BATCH", "0"))
    if forced > 0:
        return max(1, min(forced, 16, n_owned))
    if nnz * n_donor < (1 << 24):
        return max(1, min(16, n_owned))
    if not wide:
        return max(1, min(16 // n_donor, n_owned))
    costs = {}
    for R in range(1, min(n_owned, 8) + 1):
        cols = R * n_donor
        costs[R] = -(-cols // 16) * (1.25 if cols & 1 else 1.0) * (1.1 if R > 1 else 1.0) / R
    best = min(costs.values())
    return min(R for R, c in costs.items() if c <= 1.05 * best)


def _argmax_takes(best, value):
    """does ``value`` replace the best restart so far the way ``np.argmax`` over the ELBOs picks it
    (vireo_wrap.py:89-90)?  The first maximum wins; a NaN counts as the maximum (NumPy's rule), the
    first one for good -- every rank applies this to its own restarts in restart order, so the owner
    of the global argmax always holds that restart's state."""
    if best is None:
        return True
    if best[0] != best[0]:            # a NaN is kept
        return False
    return value != value or value > best[0]


class Staged:
    """raw draws of one restart, already uploaded to staging buffer ``buf`` of the runner's model"""

    def __init__(self, buf):
        self.buf = buf


class DeviceRestarts:
    """This rank's restarts on the device; the best fitted state is snapshotted device-side.
    ``template`` is a host ``Vireo`` carrying shapes, flags and priors (its own ID_prob /
    GT_prob are ignored unless passed as fixed initial values).  With ``batch`` > 1 the
    restarts are collected with ``submit`` and fitted ``batch`` at a time by one
    ``DeviceBatch`` (every sparse pass serves all of them); ``flush`` returns their ELBOs."""

    def __init__(self, counts, template, batch=None, n_owned=1):
        """batch = None: chosen by ``restart_batch`` for ``n_owned`` restarts and the stream
        words this problem uses"""
        self.counts = counts
        self.t = template
        shape = dict(n_gt=template.n_GT, learn_gt=template.learn_GT,
                     learn_theta=template.learn_theta, ase_mode=template.ASE_mode,
                     fix_beta_sum=template.fix_beta_sum)
        self.dm = DeviceModel(counts, _lib.KIND_VIREO, template.n_donor, **shape)
        template._set_device_prior(self.dm)
        if batch is None:
            info = self.dm.info()
            ad_bd = (not info["lds_cell"] and not info["lds_variant"]) or (
                info["cell_form"] == 1 and info["var_form"] in (2, 3))
            batch = restart_batch(template.n_donor, n_owned, counts.nnz, wide=ad_bd)
        self.batch = int(batch)
        self.db = None
        if self.batch > 1:
            self.db = DeviceBatch(counts, _lib.KIND_VIREO, template.n_donor, self.batch, **shape)
            template._set_device_prior(self.db)
        self.pending, self.done = [], {}
        # one restart per model: the raw draws of the next restart are uploaded (``stage``, called
        # from the thread that draws them) while the current one fits.  VIREO_STAGE_UPLOADS=0: off
        self.can_stage = (self.batch == 1 and template.n_donor <= 128 and template.n_GT <= 128
                          and os.environ.get("VIREO_STAGE_UPLOADS", "1") != "0")
        self._stage_ready = False     # the two staging buffers are reserved by the first ``stage`` call
        self._n_staged = 0
        # a staging buffer is free again once ``run`` has taken its content into the model's
        # state: the producer thread waits for that (two buffers = two slots)
        self._slots = threading.Semaphore(2)
        self._cancel = threading.Event()
        self.const = counts.binom_const()
        self.best = None            # (elbo, restart index, trace)
        self.iterations = 0

    def _theta0(self):
        t = self.t
        rows = t.n_var if t.ASE_mode else 1
        return (np.broadcast_to(t.beta_mu, (rows, t.n_GT)),
                np.broadcast_to(t.beta_sum, (rows, t.n_GT)))

    def submit(self, im, ID_raw, GT_raw, ID_fixed, GT_fixed, max_iter, delay_fit_theta):
        """Place restart ``im`` in the next slot of the batch; a full batch is fitted."""
        mu, sm = self._theta0()
        slot = len(self.pending)
        with _phase("upload+normalise"):
            if ID_fixed is not None or GT_fixed is not None:
                self.db.set_restart(slot, ID_fixed, GT_fixed, None, None, raw=False)
            self.db.set_restart(slot, ID_raw, GT_raw, mu, sm, raw=True)
            if slot == 0:
                # slots the last batch does not fill repeat its first restart (they must hold
                # a valid state; their results are dropped)
                self._fill = (ID_raw, GT_raw, ID_fixed, GT_fixed)
        self.pending.append(im)
        self._fit_args = (max_iter, delay_fit_theta)
        if len(self.pending) == self.batch:
            self._fit_pending()

    def _fit_pending(self):
        if not self.pending:
            return
        mu, sm = self._theta0()
        ID_raw, GT_raw, ID_fixed, GT_fixed = self._fill
        with _phase("upload+normalise"):
            for slot in range(len(self.pending), self.batch):
                if ID_fixed is not None or GT_fixed is not None:
                    self.db.set_restart(slot, ID_fixed, GT_fixed, None, None, raw=False)
                self.db.set_restart(slot, ID_raw, GT_raw, mu, sm, raw=True)
        max_iter, delay = self._fit_args
        with _phase("fit"):
            traces, its, _ = self.db.fit(max_iter, 5, 1e-2, delay)
        for slot, im in enumerate(self.pending):
            it = int(its[slot])
            self.iterations += it + 1
            elbo = traces[slot][:it] + self.const
            self.done[im] = elbo[-1]
            if _argmax_takes(self.best, elbo[-1]):
                with _phase("snapshot"):
                    self.db.copy_to(self.dm, slot)
                    self.dm.snapshot()
                self.best = (elbo[-1], im, elbo)
        self.pending = []
        self._fill = None

    def flush(self):
        """fit what is still pending -> {restart index: ELBO_[-1]} of everything submitted"""
        self._fit_pending()
        done, self.done = self.done, {}
        return done

    def stage(self, ID_raw, GT_raw):
        """Upload one restart's raw draws into the next staging buffer -> a token for ``run``.
        Called from the thread that produces the draws, possibly while ``run`` fits the restart
        before; at most two staged restarts may be outstanding (the producer runs one ahead)."""
        while not self._slots.acquire(timeout=0.05):
            if self._cancel.is_set():
                raise _lib.VrxError("restart search cancelled while staging an upload")
        buf = self._n_staged & 1
        self._n_staged += 1
        with _phase("stage"):
            if not self._stage_ready:
                # (two extra copies of the (ID, GT) state, ~90 MB at c3: only a search that really
                #  draws both arrays per restart pays for them)
                self.dm.stage_reserve()
                self._stage_ready = True
            self.dm.stage_raw(buf, ID_raw, GT_raw)
        return Staged(buf)

    def run(self, im, ID_raw, GT_raw, ID_fixed, GT_fixed, max_iter, delay_fit_theta):
        """Fit restart ``im`` from raw draws (normalised on the device) or, where the caller
        supplied initial values, from those (already normalised on the host).  ``ID_raw`` may be a
        ``Staged`` token (the draws are already on the device).  Returns ``ELBO_[-1]`` as
        ``Vireo.fit`` would leave it."""
        mu, sm = self._theta0()
        with _phase("upload+normalise"):
            if isinstance(ID_raw, Staged):
                self.dm.set_state_staged(ID_raw.buf, mu, sm)
                self._slots.release()
            else:
                if ID_fixed is not None or GT_fixed is not None:
                    self.dm.set_state(ID_fixed, GT_fixed, None, None)
                self.dm.set_state_raw(ID_raw, GT_raw, mu, sm)
        with _phase("fit"):
            trace, it, _ = self.dm.fit(max_iter, 5, 1e-2, delay_fit_theta)
        self.iterations += it + 1
        elbo = trace[:it] + self.const
        if _argmax_takes(self.best, elbo[-1]):
            with _phase("snapshot"):
                self.dm.snapshot()
            self.best = (elbo[-1], im, elbo)
        return elbo[-1]

    def winner(self, im, refine):
        """The host ``Vireo`` of restart ``im`` (which must be this rank's best); with
        ``refine`` the fit is continued to convergence first (vireo_wrap.py:94)."""
        if self.best is None or self.best[1] != im:
            raise _lib.VrxError("restart %d won the search but this rank kept %s" % (
                im, "nothing" if self.best is None else "restart %d" % self.best[1]))
        t = self.t
        self.dm.restore()
        t.ELBO_ = np.append(t.ELBO_, self.best[2])
        if refine:
            with _phase("final_fit"):
                trace, it, _ = self.dm.fit(200, 5, 1e-2, 0)
            self.final_iterations = it + 1
            t.ELBO_ = np.append(t.ELBO_, trace[:it] + self.const)
        with _phase("download"):
            t._pull(self.dm, want_GT=True)
        return t

    def cancel(self):
        """wake a producer thread that waits for a staging buffer (the search is being abandoned)"""
        self._cancel.set()

    def close(self):
        self._cancel.set()
        self.dm.close()
        if self.db is not None:
            self.db.close()
