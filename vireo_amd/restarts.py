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
import time

import numpy as np

from . import _lib
from .engine import DeviceModel


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


class DeviceRestarts:
    """This rank's restarts on one ``DeviceModel``; the best fitted state is snapshotted
    device-side.  ``template`` is a host ``Vireo`` carrying shapes, flags and priors (its own
    ID_prob / GT_prob are ignored unless passed as fixed initial values)."""

    def __init__(self, counts, template):
        self.counts = counts
        self.t = template
        self.dm = DeviceModel(counts, _lib.KIND_VIREO, template.n_donor, n_gt=template.n_GT,
                              learn_gt=template.learn_GT, learn_theta=template.learn_theta,
                              ase_mode=template.ASE_mode, fix_beta_sum=template.fix_beta_sum)
        template._set_device_prior(self.dm)
        self.const = counts.binom_const()
        self.best = None            # (elbo, restart index, trace)

    def run(self, im, ID_raw, GT_raw, ID_fixed, GT_fixed, max_iter, delay_fit_theta):
        """Fit restart ``im`` from raw draws (normalised on the device) or, where the caller
        supplied initial values, from those (already normalised on the host).  Returns
        ``ELBO_[-1]`` as ``Vireo.fit`` would leave it."""
        t = self.t
        rows = t.n_var if t.ASE_mode else 1
        mu = np.broadcast_to(t.beta_mu, (rows, t.n_GT))
        sm = np.broadcast_to(t.beta_sum, (rows, t.n_GT))
        with _phase("upload+normalise"):
            if ID_fixed is not None or GT_fixed is not None:
                self.dm.set_state(ID_fixed, GT_fixed, None, None)
            self.dm.set_state_raw(ID_raw, GT_raw, mu, sm)
        with _phase("fit"):
            trace, it, _ = self.dm.fit(max_iter, 5, 1e-2, delay_fit_theta)
        self.iterations = getattr(self, "iterations", 0) + it + 1
        elbo = trace[:it] + self.const
        if self.best is None or elbo[-1] > self.best[0]:     # first max wins
            with _phase("snapshot"):
                self.dm.snapshot()
            self.best = (elbo[-1], im, elbo)
        return elbo[-1]

    def winner(self, im, refine):
        """The host ``Vireo`` of restart ``im`` (which must be this rank's best); with
        ``refine`` the fit is continued to convergence first (vireo_wrap.py:94)."""
        assert self.best is not None and self.best[1] == im
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

    def close(self):
        self.dm.close()
