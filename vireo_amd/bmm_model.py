"""``BinomMixtureVB``: binomial mixture (clone mode), drop-in for the reference class
(vireoSNP/utils/bmm_model.py:9-263) with the VB loop on MI355X.  theta is a Beta per
(variant, clone); there is no genotype layer.  No CPU fallback.
"""
import numpy as np

from . import _lib
from .counts import device_counts
from .dist import LocalComm, first_record, gather_restart_elbos, my_restarts
from .engine import DeviceBatch, DeviceModel
from .restarts import LegacyStream, restart_batch
from .vireo_base import normalize


def _is_record(seen, value):
    """``i == 0 or ELBO_iters[-1] > np.max(ELBO_inits[:-1])`` (bmm_model.py:248), NaNs as NumPy
    treats them: a NaN among the earlier values makes the comparison False for good"""
    return not seen or bool(value > np.max(seen))


class BinomMixtureVB():
    """Key properties: beta_mu, beta_sum (n_var, n_donor); ID_prob (n_cell, n_donor);
    ELBO_iters (trace of the best initialisation + final fit); ELBO_inits."""

    def __init__(self, n_cell, n_var, n_donor, fix_beta_sum=False,
                 beta_mu_init=None, beta_sum_init=None, ID_prob_init=None):
        # bmm_model.py:49-63
        self.n_var = n_var
        self.n_cell = n_cell
        self.n_donor = n_donor
        self.fix_beta_sum = fix_beta_sum
        self.ID_prob_init = ID_prob_init
        self.beta_mu_init = beta_mu_init
        self.beta_sum_init = beta_sum_init
        self.set_prior()
        self.set_initial(self.beta_mu_init, self.beta_sum_init, self.ID_prob_init)

    def set_initial(self, beta_mu_init=None, beta_sum_init=None, ID_prob_init=None):
        """bmm_model.py:65-85: mu = 0.5, sum = 30, ID_prob = normalised rand(M, K)."""
        shape = (self.n_var, self.n_donor)
        self.beta_mu = np.ones(shape) * 0.5 if beta_mu_init is None else beta_mu_init
        self.beta_sum = (np.ones(self.beta_mu.shape) * 30 if beta_sum_init is None
                         else beta_sum_init)
        if ID_prob_init is None:
            self.ID_prob = normalize(np.random.rand(self.n_cell, self.n_donor))
        else:
            self.ID_prob = normalize(ID_prob_init, axis=1)
        self.ELBO_iters = np.array([])

    def set_prior(self, ID_prior=None, beta_mu_prior=None, beta_sum_prior=None):
        """bmm_model.py:87-105: Beta(1, 1) on theta, uniform ID prior."""
        if beta_mu_prior is None:
            beta_mu_prior = np.ones((self.n_var, self.n_donor)) * 0.5
        if beta_sum_prior is None:
            beta_sum_prior = np.ones(beta_mu_prior.shape) * 2.0
        self.theta_s1_prior = beta_mu_prior * beta_sum_prior
        self.theta_s2_prior = (1 - beta_mu_prior) * beta_sum_prior
        if ID_prior is None:
            self.ID_prior = normalize(np.ones((self.n_cell, self.n_donor)))
        else:
            self.ID_prior = ID_prior[None, :] if len(ID_prior.shape) == 1 else ID_prior

    @property
    def theta_s1(self):
        return self.beta_mu * self.beta_sum

    @property
    def theta_s2(self):
        return (1 - self.beta_mu) * self.beta_sum

    # ------------------------------------------------------------------ device plumbing
    def _device_model(self, AD, DP):
        counts = device_counts(AD, DP)
        if counts.shape != (self.n_var, self.n_cell):
            raise ValueError("AD/DP have shape %s but the model was built for (%d, %d)"
                             % (counts.shape, self.n_var, self.n_cell))
        dm = DeviceModel(counts, _lib.KIND_BMM, self.n_donor, fix_beta_sum=self.fix_beta_sum)
        self._push(dm)
        return dm

    def _push(self, dm):
        shape = (self.n_var, self.n_donor)
        dm.set_state(self.ID_prob, None, np.broadcast_to(self.beta_mu, shape),
                     np.broadcast_to(self.beta_sum, shape))
        self._push_prior(dm)

    def _push_prior(self, dm):
        # a constant theta prior travels as one broadcast row
        s1, s2 = self.theta_s1_prior, self.theta_s2_prior
        if np.all(s1 == s1.flat[0]) and np.all(s2 == s2.flat[0]):
            s1 = np.full((1, self.n_donor), s1.flat[0])
            s2 = np.full((1, self.n_donor), s2.flat[0])
            # the kernel indexes a 1-row prior at element 0 only
        dm.set_prior(self.ID_prior, None, s1, s2)

    def _pull(self, dm):
        self.ID_prob, _, self.beta_mu, self.beta_sum = dm.get_state()

    # ------------------------------------------------------------------ single updates
    def get_E_logLik(self, AD, DP):
        """E_theta[log P(AD | DP, theta, Z)], (n_cell, n_donor) (bmm_model.py:118-130)."""
        dm = self._device_model(AD, DP)
        dm.step(_lib.STEP_LOGLIK)
        L = dm.get_loglik()
        dm.close()
        return L

    def update_theta_size(self, AD, DP):
        """bmm_model.py:133-144."""
        dm = self._device_model(AD, DP)
        dm.step(_lib.STEP_THETA)
        _, _, self.beta_mu, self.beta_sum = dm.get_state()
        dm.close()

    def update_ID_prob(self, AD=None, DP=None, logLik_ID=None):
        """bmm_model.py:147-154.  With logLik_ID given this is a row softmax of
        logLik_ID + log(ID_prior); it still runs on the device and therefore needs AD, DP
        to find the device problem."""
        if AD is None or DP is None:
            raise ValueError("vireo_amd: update_ID_prob needs AD and DP")
        dm = self._device_model(AD, DP)
        if logLik_ID is None:
            dm.step(_lib.STEP_ID)
        else:
            dm.set_loglik(logLik_ID)
            dm.step(_lib.STEP_SOFTMAX)
        self.ID_prob = dm.get_state()[0]
        dm.close()

    def get_ELBO(self, AD=None, DP=None, logLik_ID=None):
        """bmm_model.py:157-175 (LB_p - KL_ID - KL_theta)."""
        if AD is None or DP is None:
            raise ValueError("vireo_amd: get_ELBO needs AD and DP")
        dm = self._device_model(AD, DP)
        if logLik_ID is None:
            dm.step(_lib.STEP_LOGLIK)
        else:
            dm.set_loglik(logLik_ID)
        val = dm.step(_lib.STEP_ELBO)
        dm.close()
        return val

    # ------------------------------------------------------------------ the VB loop
    def _fit_BV(self, AD, DP, max_iter=200, min_iter=20, epsilon_conv=1e-2,
                verbose=True, _dm=None):
        """bmm_model.py:178-201 on the GPU; appends ELBO[:it] to ELBO_iters."""
        dm = _dm if _dm is not None else self._device_model(AD, DP)
        if _dm is not None:
            self._push(dm)
        trace, it, _flags = dm.fit(max_iter, min_iter, epsilon_conv)
        self._pull(dm)
        if _dm is None:
            dm.close()
        if verbose:
            self._warn(trace, it, min_iter, max_iter)
        self.ELBO_iters = np.append(self.ELBO_iters, trace[:it])

    @staticmethod
    def _warn(trace, it, min_iter, max_iter):
        """the reference's prints (bmm_model.py:192-197)"""
        for i in range(min_iter + 1, it + 1):
            if trace[i] - trace[i - 1] < -1e-6:
                print("Warning: ELBO decreases %.8f to %.8f!\n" % (trace[i - 1], trace[i]))
            elif i == max_iter - 1:
                print("Warning: VB did not converge!\n")

    def _draw_initial(self):
        """one ``set_initial`` of the n_init loop (bmm_model.py:243-245) -> the state it leaves"""
        shape = (self.n_var, self.n_donor)
        self.set_initial(self.beta_mu_init, self.beta_sum_init, self.ID_prob_init)
        return (self.ID_prob, np.broadcast_to(self.beta_mu, shape),
                np.broadcast_to(self.beta_sum, shape))

    def _skip_initials(self, n):
        """pass over the draws of ``n`` initialisations that other ranks fit: the global legacy
        stream moves exactly as n ``set_initial`` calls would move it (rand(n_cell, n_donor) each,
        bmm_model.py:81-82; nothing when ID_prob_init is given)"""
        if n > 0 and self.ID_prob_init is None:
            LegacyStream().skip(n * self.n_cell * self.n_donor)

    def _fit_inits_batched(self, counts, dm, n_init, mine, R, max_iter_pre, min_iter=20,
                           epsilon_conv=1e-2, verbose=True):
        """This rank's share ``mine`` of the n_init short fits of ``fit`` (bmm_model.py:241-252),
        R at a time in one device model (vrx_model_cfg.n_batch): same draws in the same order,
        same prints, the best state stays on the device (a snapshot of ``dm``).
        -> ({restart: ELBO}, (ELBO, restart, trace) of the first maximum or None)"""
        db = DeviceBatch(counts, _lib.KIND_BMM, self.n_donor, R, fix_beta_sum=self.fix_beta_sum)
        self._push_prior(db)
        local, best, seen = {}, None, []
        consumed = 0
        for base in range(0, len(mine), R):
            ids = mine[base:base + R]
            first = None
            for slot in range(R):
                if slot < len(ids):
                    self._skip_initials(ids[slot] - consumed)
                    consumed = ids[slot] + 1
                    state = self._draw_initial()
                    first = first or state
                else:
                    state = first          # idle slots repeat the batch's first restart
                db.set_restart(slot, state[0], None, state[1], state[2])
            traces, its, _ = db.fit(max_iter_pre, min_iter, epsilon_conv)
            for slot, i in enumerate(ids):
                trace, it = traces[slot], int(its[slot])
                if verbose:
                    self._warn(trace, it, min_iter, max_iter_pre)
                local[i] = trace[:it][-1]
                if _is_record(seen, local[i]):
                    db.copy_to(dm, slot)
                    dm.snapshot()
                    best = (local[i], i, trace[:it] + 0)
                seen.append(local[i])
        self._skip_initials(n_init - consumed)
        db.close()
        return local, best

    def _fit_inits_single(self, counts, dm, n_init, mine, max_iter_pre, **kwargs):
        """the same, one restart per device model (problems with no idle columns)"""
        local, best, seen = {}, None, []
        consumed = 0
        for i in mine:
            self._skip_initials(i - consumed)
            consumed = i + 1
            self.set_initial(self.beta_mu_init, self.beta_sum_init, self.ID_prob_init)
            self._fit_BV(counts, None, max_iter=max_iter_pre, _dm=dm, **kwargs)
            local[i] = self.ELBO_iters[-1]
            if _is_record(seen, local[i]):
                dm.snapshot()
                best = (local[i], i, self.ELBO_iters + 0)
            seen.append(local[i])
        self._skip_initials(n_init - consumed)
        return local, best

    def fit(self, AD, DP, n_init=10, max_iter=200, max_iter_pre=100,
            random_seed=None, comm=None, **kwargs):
        """VB with multiple initialisations (bmm_model.py:204-263): n_init short fits
        (max_iter_pre), keep the best by ELBO_iters[-1] (strictly greater = the first maximum),
        re-fit it (max_iter), add the binomial-coefficient constant.  kwargs -> _fit_BV
        (min_iter=20, epsilon_conv=1e-2, verbose=True).

        ``comm`` (vireo_amd/dist.py) shards the initialisations like ``vireo_wrap``'s restarts
        (SURVEY.md 8e): initialisation i on rank i % world -- every rank walks the whole random
        stream and forms only its own draws --, one all-gather of the n_init ELBOs, the owner of
        the first maximum runs the final fit and broadcasts the state; every rank returns with the
        same attributes, bit for bit those of the world-1 call."""
        comm = LocalComm() if comm is None else comm
        if random_seed is not None:
            np.random.seed(random_seed)
        counts = device_counts(AD, DP)
        const = counts.binom_const()
        dm = self._device_model(counts, None)
        mine = my_restarts(n_init, comm.rank, comm.world)
        R = restart_batch(self.n_donor, len(mine), counts.nnz, wide=False)
        if R > 1:
            local, best = self._fit_inits_batched(counts, dm, n_init, mine, R, max_iter_pre, **kwargs)
        else:
            local, best = self._fit_inits_single(counts, dm, n_init, mine, max_iter_pre, **kwargs)
        elbo_inits = gather_restart_elbos(comm, n_init, local)
        winner = first_record(elbo_inits)             # bmm_model.py:248-252, NaN rule included
        owner = winner % comm.world
        if comm.rank == owner:
            if best is None or best[1] != winner:
                # (only a NaN ELBO on ANOTHER rank can do this: the owner applied the reference's
                #  rule to its own initialisations and could not see that an earlier NaN elsewhere
                #  had frozen the choice; at world 1 the two always agree)
                raise _lib.VrxError("initialisation %d is the one bmm_model.py:248 keeps, but rank %d "
                                    "kept %s (a NaN ELBO among the initialisations: %s)" % (
                                        winner, comm.rank, "none" if best is None else best[1], elbo_inits))
            dm.restore()
            self._pull(dm)
            self.set_initial(self.beta_mu, self.beta_sum, self.ID_prob)
            self.ELBO_iters = best[2]
            self._fit_BV(counts, None, max_iter=max_iter, _dm=dm, **kwargs)
        if comm.world > 1 and hasattr(comm, "bcast_model"):      # device to device under RCCL
            comm.bcast_model(dm, owner)
            if comm.rank != owner:
                self._pull(dm)
        elif comm.world > 1:
            for name in ("ID_prob", "beta_mu", "beta_sum"):
                setattr(self, name, comm.bcast(getattr(self, name), owner))
        dm.close()
        if comm.world > 1:
            n = comm.bcast(np.array([float(len(self.ELBO_iters))]), owner)
            trace = self.ELBO_iters if comm.rank == owner else np.zeros(int(n[0]))
            self.ELBO_iters = comm.bcast(trace, owner)
        self.ELBO_iters = self.ELBO_iters + const
        self.ELBO_inits = np.array(elbo_inits) + const
