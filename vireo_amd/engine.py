"""Thin Python handle on a device-resident model (C handle ``vrx_model``).

Everything numeric happens in libvireo_hip.so; this class only moves the reference's
NumPy attributes across the C ABI.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import dptr, f64


def _prior_rows(P, full_rows):
    """(array-or-None, rows) for a prior table: 0 = uniform (constant table), 1 = one
    broadcast row, full_rows = per-row table."""
    if P is None:
        return None, 0
    P = np.asarray(P, dtype=np.float64)
    if P.size and P.min() == P.max():   # constant = uniform after normalisation
        return None, 0
    if P.shape[0] == 1:
        return f64(P), 1
    if P.shape[0] != full_rows:
        raise ValueError("prior with %d rows does not match %d" % (P.shape[0], full_rows))
    return f64(P), full_rows


class DeviceModel:
    def __init__(self, counts, kind, n_donor, n_gt=3, learn_gt=True, learn_theta=True,
                 ase_mode=False, fix_beta_sum=False, n_batch=1):
        self.counts = counts            # keeps the vrx_problem alive
        self.R = int(n_batch)
        self.kind = kind
        self.K, self.T = int(n_donor), int(n_gt)
        self.N, self.M = counts.n_var, counts.n_cell
        if kind == _lib.KIND_VIREO:
            self.theta_shape = (self.N if ase_mode else 1, self.T)
        else:
            self.theta_shape = (self.N, self.K)
        cfg = _lib.ModelCfg(kind=kind, n_donor=self.K, n_gt=self.T, learn_gt=int(bool(learn_gt)),
                            learn_theta=int(bool(learn_theta)), ase_mode=int(bool(ase_mode)),
                            fix_beta_sum=int(bool(fix_beta_sum)), n_batch=self.R)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().vrx_model_create(counts.handle, C.byref(cfg), C.byref(self._h)))
        self._fin = weakref.finalize(self, _lib.lib().vrx_model_destroy, self._h)

    def close(self):
        self._fin()

    # ---- state / priors ---------------------------------------------------------------
    def _check(self, a, shape, name):
        if a is None:
            return None
        a = f64(a)
        if a.shape != tuple(shape):
            raise ValueError("%s has shape %s, expected %s" % (name, a.shape, tuple(shape)))
        return a

    def set_state(self, ID_prob=None, GT_prob=None, beta_mu=None, beta_sum=None):
        ID_prob = self._check(ID_prob, (self.M, self.K), "ID_prob")
        if self.kind == _lib.KIND_VIREO:
            GT_prob = self._check(GT_prob, (self.N, self.K, self.T), "GT_prob")
        else:
            GT_prob = None
        beta_mu = self._check(beta_mu, self.theta_shape, "beta_mu")
        beta_sum = self._check(beta_sum, self.theta_shape, "beta_sum")
        _lib.check(_lib.lib().vrx_model_set_state(self._h, dptr(ID_prob), dptr(GT_prob),
                                                  dptr(beta_mu), dptr(beta_sum)))

    def set_state_raw(self, ID_raw=None, GT_raw=None, beta_mu=None, beta_sum=None):
        """upload un-normalised draws; normalised over the last axis on the device exactly
        like ``normalize`` does on the host (vrx_model_set_state_raw)"""
        ID_raw = self._check(ID_raw, (self.M, self.K), "ID_raw")
        GT_raw = (self._check(GT_raw, (self.N, self.K, self.T), "GT_raw")
                  if self.kind == _lib.KIND_VIREO else None)
        beta_mu = self._check(beta_mu, self.theta_shape, "beta_mu")
        beta_sum = self._check(beta_sum, self.theta_shape, "beta_sum")
        _lib.check(_lib.lib().vrx_model_set_state_raw(self._h, dptr(ID_raw), dptr(GT_raw),
                                                      dptr(beta_mu), dptr(beta_sum)))

    # staged uploads: the next restart's raw draws go up while this one fits
    def stage_reserve(self):
        _lib.check(_lib.lib().vrx_model_stage_reserve(self._h))

    def stage_raw(self, buf, ID_raw, GT_raw):
        """staging buffer ``buf`` (0 | 1) <- raw draws; thread-safe against a running ``fit``"""
        ID_raw = self._check(ID_raw, (self.M, self.K), "ID_raw")
        GT_raw = self._check(GT_raw, (self.N, self.K, self.T), "GT_raw")
        _lib.check(_lib.lib().vrx_model_stage_raw(self._h, int(buf), dptr(ID_raw), dptr(GT_raw)))

    def set_state_staged(self, buf, beta_mu=None, beta_sum=None):
        beta_mu = self._check(beta_mu, self.theta_shape, "beta_mu")
        beta_sum = self._check(beta_sum, self.theta_shape, "beta_sum")
        _lib.check(_lib.lib().vrx_model_set_state_staged(self._h, int(buf), dptr(beta_mu), dptr(beta_sum)))

    def snapshot(self):
        """keep the current state in a device-side slot"""
        _lib.check(_lib.lib().vrx_model_snapshot(self._h, 0))

    def restore(self):
        _lib.check(_lib.lib().vrx_model_snapshot(self._h, 1))

    def get_state(self, want_GT=True):
        ID = np.empty((self.M, self.K))
        GT = np.empty((self.N, self.K, self.T)) if (want_GT and self.kind == _lib.KIND_VIREO) else None
        mu = np.empty(self.theta_shape)
        sm = np.empty(self.theta_shape)
        _lib.check(_lib.lib().vrx_model_get_state(self._h, dptr(ID), dptr(GT), dptr(mu), dptr(sm)))
        return ID, GT, mu, sm

    def set_prior(self, ID_prior, GT_prior, theta_s1_prior, theta_s2_prior):
        idp, id_rows = _prior_rows(ID_prior, self.M)
        if self.kind == _lib.KIND_VIREO:
            gtp, gt_rows = _prior_rows(GT_prior, self.N)
            if gtp is not None and gtp.shape[1:] != (self.K, self.T):
                raise ValueError("GT_prior has shape %s" % (gtp.shape,))
        else:
            gtp, gt_rows = None, 0
        if idp is not None and idp.shape[1] != self.K:
            raise ValueError("ID_prior has shape %s" % (idp.shape,))
        s1, s2 = f64(theta_s1_prior), f64(theta_s2_prior)
        if self.kind == _lib.KIND_BMM and s1.ndim == 2 and s1.shape[0] == 1 and not (
                np.all(s1 == s1.flat[0]) and np.all(s2 == s2.flat[0])):
            # a (1, n_donor) prior that differs per clone broadcasts over the variants like in
            # the reference (bmm_model.py:92-98); the kernel reads a 1-row prior at [0] only
            s1 = f64(np.broadcast_to(s1, self.theta_shape))
            s2 = f64(np.broadcast_to(s2, self.theta_shape))
        if s1.shape != s2.shape or s1.ndim != 2 or s1.shape[1] != self.theta_shape[1] \
                or s1.shape[0] not in (1, self.theta_shape[0]):
            raise ValueError("theta prior has shape %s" % (s1.shape,))
        _lib.check(_lib.lib().vrx_model_set_prior(self._h, dptr(idp), id_rows, dptr(gtp), gt_rows,
                                                  dptr(s1), dptr(s2), s1.shape[0]))

    # ---- compute ----------------------------------------------------------------------
    def fit(self, max_iter, min_iter, epsilon_conv, delay_fit_theta=0):
        """-> (every computed ELBO [it+1 values, no binomial constant], it, warn_flags)"""
        trace = np.zeros(max_iter)
        it = C.c_int32(0)
        flags = C.c_int32(0)
        _lib.check(_lib.lib().vrx_model_fit(self._h, int(max_iter), int(min_iter),
                                            float(epsilon_conv), int(delay_fit_theta),
                                            dptr(trace), C.byref(it), C.byref(flags)))
        return trace[:it.value + 1], it.value, flags.value

    def step(self, which):
        out = C.c_double(0.0)
        _lib.check(_lib.lib().vrx_model_step(self._h, which, C.byref(out)))
        return out.value

    def get_loglik(self):
        L = np.empty((self.M, self.K))
        _lib.check(_lib.lib().vrx_model_get_loglik(self._h, dptr(L)))
        return L

    def set_loglik(self, L):
        L = self._check(L, (self.M, self.K), "logLik_ID")
        _lib.check(_lib.lib().vrx_model_set_loglik(self._h, dptr(L)))

    def elbo_parts(self):
        p = np.zeros(4)
        _lib.check(_lib.lib().vrx_model_get_elbo_parts(self._h, dptr(p)))
        return p

    def info(self):
        """which kernels / formats / tilings this model's passes use (vrx_model_info)"""
        a = np.zeros(16, dtype=np.int32)
        _lib.check(_lib.lib().vrx_model_info(self._h, a.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(lds_variant=bool(a[0]), lds_cell=bool(a[1]), fmt_variant=int(a[2]),
                    fmt_cell=int(a[3]), tiles_variant=int(a[4]), tiles_cell=int(a[5]),
                    ranges_variant=int(a[6]), ranges_cell=int(a[7]),
                    pad_variant=a[8] / 1000.0, pad_cell=a[9] / 1000.0,
                    extra_pieces_variant=int(a[10]), extra_pieces_cell=int(a[11]),
                    cell_form=int(a[12]), var_form=int(a[13]), n_batch=int(a[14]),
                    imbalance_variant=(int(a[15]) & 0xffff) / 1000.0,
                    imbalance_cell=(int(a[15]) >> 16) / 1000.0)

    # ---- timing -----------------------------------------------------------------------
    def profile(self, enable=True):
        _lib.check(_lib.lib().vrx_model_profile(self._h, int(enable)))

    def profile_read(self):
        ms = np.zeros(_lib.KERN_COUNT)
        n = np.zeros(_lib.KERN_COUNT, dtype=np.int64)
        _lib.check(_lib.lib().vrx_model_profile_read(self._h, dptr(ms),
                                                     n.ctypes.data_as(C.POINTER(C.c_int64))))
        return ms, n

    def run_iters(self, n_iter, theta_from_iter=0):
        """n_iter iterations back to back, no convergence test -> (elbo trace, wall ms)"""
        trace = np.zeros(n_iter)
        ms = C.c_double(0.0)
        _lib.check(_lib.lib().vrx_model_run_iters(self._h, int(n_iter), int(theta_from_iter),
                                                  dptr(trace), C.byref(ms)))
        return trace, ms.value


class DeviceBatch(DeviceModel):
    """``n_batch`` restarts of one model shape in ONE device model (vrx_model_cfg.n_batch):
    every sparse pass serves all of them, each keeps its own theta, ELBO trace and stop rule.
    Slots are filled one restart at a time in the single-model layouts; the winner moves to a
    single ``DeviceModel`` on the device."""

    def __init__(self, counts, kind, n_donor, n_batch, **kw):
        if not 1 <= int(n_batch) <= 16:
            raise ValueError("n_batch must be in 1..16")
        super().__init__(counts, kind, n_donor, n_batch=n_batch, **kw)

    def _single_only(self, *a, **k):
        raise TypeError("not available on a restart batch; use set_restart / copy_to")

    set_state = set_state_raw = get_state = snapshot = restore = _single_only
    stage_reserve = stage_raw = set_state_staged = _single_only
    get_loglik = set_loglik = _single_only

    def set_restart(self, r, ID=None, GT=None, beta_mu=None, beta_sum=None, raw=False):
        ID = self._check(ID, (self.M, self.K), "ID")
        GT = (self._check(GT, (self.N, self.K, self.T), "GT")
              if self.kind == _lib.KIND_VIREO else None)
        beta_mu = self._check(beta_mu, self.theta_shape, "beta_mu")
        beta_sum = self._check(beta_sum, self.theta_shape, "beta_sum")
        _lib.check(_lib.lib().vrx_model_set_restart(self._h, int(r), dptr(ID), dptr(GT),
                                                    dptr(beta_mu), dptr(beta_sum), int(bool(raw))))

    def copy_to(self, single, r):
        """``single`` (a DeviceModel of the same problem and shape) <- slot r, on the device"""
        _lib.check(_lib.lib().vrx_model_copy_restart(single._h, self._h, int(r)))

    def fit(self, max_iter, min_iter, epsilon_conv, delay_fit_theta=0):
        """-> per restart: (list of traces [it_r + 1 values each], it [R], warn_flags [R])"""
        trace = np.zeros((self.R, max_iter))
        it = np.zeros(self.R, dtype=np.int32)
        flags = np.zeros(self.R, dtype=np.int32)
        i32 = C.POINTER(C.c_int32)
        _lib.check(_lib.lib().vrx_model_fit(self._h, int(max_iter), int(min_iter),
                                            float(epsilon_conv), int(delay_fit_theta),
                                            dptr(trace), it.ctypes.data_as(i32),
                                            flags.ctypes.data_as(i32)))
        return [trace[r, :it[r] + 1] for r in range(self.R)], it, flags

    def step(self, which):
        out = np.zeros(self.R)
        _lib.check(_lib.lib().vrx_model_step(self._h, which, dptr(out)))
        return out

    def elbo_parts(self):
        p = np.zeros((self.R, 4))
        _lib.check(_lib.lib().vrx_model_get_elbo_parts(self._h, dptr(p)))
        return p

    def run_iters(self, n_iter, theta_from_iter=0):
        trace = np.zeros((self.R, n_iter))
        ms = C.c_double(0.0)
        _lib.check(_lib.lib().vrx_model_run_iters(self._h, int(n_iter), int(theta_from_iter),
                                                  dptr(trace), C.byref(ms)))
        return trace, ms.value
