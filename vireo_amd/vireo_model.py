"""``Vireo``: the donor-deconvolution VB model, drop-in for the reference class
(vireoSNP/utils/vireo_model.py:11-315) with the coordinate-ascent loop on MI355X.

Same constructor, same public attributes (plain writable ndarrays), same ``fit()``
semantics (warm restart, ELBO_ growth, the ELBO[:it] truncation).  The initial state and
priors are prepared on the host with NumPy so that the legacy global RNG stream is
consumed exactly as the reference does; every per-iteration update runs in HIP kernels
behind the C ABI (include/vireo_hip.h).  No CPU fallback.
"""
import numpy as np

from . import _lib
from .counts import device_counts
from .engine import DeviceModel
from .vireo_base import normalize


class Vireo():
    """Variational inference for reconstruction of ensemble origin.

    Key properties (as in the reference)
    ------------------------------------
    beta_mu, beta_sum : (1, n_GT) or (n_var, n_GT) in ASE mode -- Beta posterior of theta
    ID_prob : (n_cell, n_donor) -- posterior cell assignment
    GT_prob : (n_var, n_donor, n_GT) -- posterior genotype; ``GP_prob`` is an alias
    ELBO_   : 1-D trace, grows across fit() calls
    """

    def __init__(self, n_cell, n_var, n_donor, n_GT=3, learn_GT=True,
                 learn_theta=True, ASE_mode=False, fix_beta_sum=False,
                 beta_mu_init=None, beta_sum_init=None, ID_prob_init=None,
                 GT_prob_init=None):
        # vireo_model.py:61-76
        self.n_GT = n_GT
        self.n_var = n_var
        self.n_cell = n_cell
        self.n_donor = n_donor
        self.learn_GT = learn_GT
        self.ASE_mode = ASE_mode
        self.learn_theta = learn_theta
        self.fix_beta_sum = fix_beta_sum
        self.ELBO_ = np.zeros((0))
        self.set_initial(beta_mu_init, beta_sum_init, ID_prob_init, GT_prob_init)
        self.set_prior()

    # ------------------------------------------------------------------ host-side setup
    def set_initial(self, beta_mu_init=None, beta_sum_init=None,
                    ID_prob_init=None, GT_prob_init=None):
        """Initial values (vireo_model.py:78-104).  RNG order: rand(n_cell, n_donor) first,
        then rand(n_var, n_donor, n_GT) -- from the global legacy NumPy stream."""
        rows = self.n_var if self.ASE_mode else 1
        if beta_mu_init is None:
            grid = np.linspace(0.01, 0.99, self.n_GT).reshape(1, -1)
            self.beta_mu = np.ones((rows, self.n_GT)) * grid
        else:
            self.beta_mu = beta_mu_init
        if beta_sum_init is None:
            self.beta_sum = np.ones((rows, self.n_GT)) * 50
        else:
            self.beta_sum = beta_sum_init
        if ID_prob_init is None:
            self.ID_prob = normalize(np.random.rand(self.n_cell, self.n_donor))
        else:
            self.ID_prob = normalize(ID_prob_init, axis=1)
        if GT_prob_init is None:
            self.GT_prob = normalize(np.random.rand(self.n_var, self.n_donor, self.n_GT))
        else:
            self.GT_prob = normalize(GT_prob_init)

    def set_prior(self, GT_prior=None, ID_prior=None, beta_mu_prior=None,
                  beta_sum_prior=None, min_GP=0.00001):
        """Priors in the shapes of their variables (vireo_model.py:107-137).  Like the
        reference, a passed GT_prior is clipped to [min_GP, 1-min_GP] IN PLACE."""
        if beta_mu_prior is None:
            beta_mu_prior = np.linspace(0.01, 0.99, self.beta_mu.shape[1])[None, :]
        if beta_sum_prior is None:
            beta_sum_prior = np.ones(beta_mu_prior.shape) * 50.0
        self.theta_s1_prior = beta_mu_prior * beta_sum_prior
        self.theta_s2_prior = (1 - beta_mu_prior) * beta_sum_prior

        if ID_prior is None:
            self.ID_prior = None          # uniform; the array is formed when somebody reads it
        else:
            self.ID_prior = ID_prior[None, :] if len(ID_prior.shape) == 1 else ID_prior

        if GT_prior is None:
            self.GT_prior = None          # uniform, likewise
        else:
            if len(GT_prior.shape) == 2:
                GT_prior = GT_prior[None, :, :]
            GT_prior[GT_prior < min_GP] = min_GP
            GT_prior[GT_prior > 1 - min_GP] = 1 - min_GP
            self.GT_prior = normalize(GT_prior)

    # The default priors are uniform (vireo_model.py:120,127: normalize(np.ones(shape))).  The
    # reference materialises them for every model -- 38 MB at c3 -- and the device would then
    # read a log-prior table on every iteration; here "uniform" is kept as None until host code
    # reads the attribute, and the device uses its scalar-prior kernels.
    @property
    def ID_prior(self):
        if self._ID_prior is None:
            self._ID_prior = normalize(np.ones((self.n_cell, self.n_donor)))
        return self._ID_prior

    @ID_prior.setter
    def ID_prior(self, value):
        self._ID_prior = value

    @property
    def GT_prior(self):
        if self._GT_prior is None:
            self._GT_prior = normalize(np.ones((self.n_var, self.n_donor, self.n_GT)))
        return self._GT_prior

    @GT_prior.setter
    def GT_prior(self, value):
        self._GT_prior = value

    def __getstate__(self):
        """plain NumPy state only (the reference ships models through multiprocessing.Pool,
        vireo_wrap.py:74-83): the handle on the device problem stays behind"""
        state = dict(self.__dict__)
        state.pop("_last_counts", None)
        return state

    @property
    def GP_prob(self):
        """pre-0.2.2 name of GT_prob (doc/release.rst:101)."""
        return self.GT_prob

    @property
    def theta_s1(self):
        """Beta concentration-1 of theta's posterior (vireo_model.py:139-142)."""
        return self.beta_mu * self.beta_sum

    @property
    def theta_s2(self):
        """Beta concentration-2 of theta's posterior (vireo_model.py:144-147)."""
        return (1 - self.beta_mu) * self.beta_sum

    # ------------------------------------------------------------------ device plumbing
    def _device_model(self, AD, DP):
        counts = device_counts(AD, DP)
        if counts.shape != (self.n_var, self.n_cell):
            raise ValueError("AD/DP have shape %s but the model was built for (%d, %d)"
                             % (counts.shape, self.n_var, self.n_cell))
        self._last_counts = counts
        dm = DeviceModel(counts, _lib.KIND_VIREO, self.n_donor, n_gt=self.n_GT,
                         learn_gt=self.learn_GT, learn_theta=self.learn_theta,
                         ase_mode=self.ASE_mode, fix_beta_sum=self.fix_beta_sum)
        rows = self.n_var if self.ASE_mode else 1
        shape = (rows, self.n_GT)
        dm.set_state(self.ID_prob, self.GT_prob, np.broadcast_to(self.beta_mu, shape),
                     np.broadcast_to(self.beta_sum, shape))
        self._set_device_prior(dm)
        return dm, counts

    def _set_device_prior(self, dm):
        GT_prior = self._GT_prior         # None: uniform (never materialised)
        full = (self.n_var, self.n_donor, self.n_GT)
        if GT_prior is not None and GT_prior.shape[0] != 1 and GT_prior.shape != full:
            GT_prior = np.broadcast_to(GT_prior, full)
        dm.set_prior(self._ID_prior, GT_prior, self.theta_s1_prior, self.theta_s2_prior)

    def _pull(self, dm, want_GT=True):
        ID, GT, mu, sm = dm.get_state(want_GT=want_GT)
        self.ID_prob = ID
        if GT is not None:
            self.GT_prob = GT
        self.beta_mu, self.beta_sum = mu, sm

    # ------------------------------------------------------------------ single updates
    def update_theta_size(self, AD, DP):
        """Coordinate ascent for theta's Beta posterior (vireo_model.py:165-185)."""
        dm, _ = self._device_model(AD, DP)
        dm.step(_lib.STEP_THETA)
        _, _, self.beta_mu, self.beta_sum = dm.get_state(want_GT=False)
        dm.close()

    def update_ID_prob(self, AD, DP):
        """Coordinate ascent for the assignment posterior; returns logLik_ID
        (vireo_model.py:187-201)."""
        dm, _ = self._device_model(AD, DP)
        dm.step(_lib.STEP_ID)
        self.ID_prob = dm.get_state(want_GT=False)[0]
        L = dm.get_loglik()
        dm.close()
        return L

    def update_GT_prob(self, AD, DP):
        """Coordinate ascent for the genotype posterior (vireo_model.py:204-219)."""
        dm, _ = self._device_model(AD, DP)
        dm.step(_lib.STEP_GT)
        self.GT_prob = dm.get_state()[1]
        dm.close()

    def get_ELBO(self, logLik_ID, AD=None, DP=None):
        """Evidence lower bound of the current parameters (vireo_model.py:222-248).
        logLik_ID None -> recomputed from AD, DP (:227-234)."""
        if AD is None:          # like the reference: get_ELBO(logLik_ID) right after an update
            AD = getattr(self, "_last_counts", None)
            if AD is None:
                raise ValueError("vireo_amd: get_ELBO needs AD and DP (no earlier fit/update "
                                 "call on this object to take the device problem from)")
        dm, _ = self._device_model(AD, DP)
        if logLik_ID is None:
            dm.step(_lib.STEP_LOGLIK)
        else:
            dm.set_loglik(logLik_ID)
        val = dm.step(_lib.STEP_ELBO)
        dm.close()
        return val

    # ------------------------------------------------------------------ the VB loop
    def _fit_VB(self, AD, DP, max_iter=200, min_iter=5, epsilon_conv=1e-2,
                delay_fit_theta=0, verbose=True):
        """vireo_model.py:251-276 on the GPU; returns ELBO[:it] like the reference
        (the last computed value is dropped)."""
        dm, _ = self._device_model(AD, DP)
        trace, it, _flags = dm.fit(max_iter, min_iter, epsilon_conv, delay_fit_theta)
        self._pull(dm, want_GT=True)
        dm.close()
        if verbose:          # replay the reference's prints (vireo_model.py:266-272)
            for i in range(min_iter + 1, it + 1):
                if trace[i] < trace[i - 1] - 1e-6:
                    print("Warning: Lower bound decreases!\n")
                elif i == max_iter - 1:
                    print("Warning: VB did not converge!\n")
        return trace[:it]

    def fit(self, AD, DP, max_iter=200, min_iter=5, epsilon_conv=1e-2,
            delay_fit_theta=0, verbose=True, n_inits=50, nproc=1):
        """Fit with coordinate ascent (vireo_model.py:278-315).

        AD, DP : scipy.sparse (CSC or CSR, int or float counts) or dense ndarray,
                 (n_var, n_cell); or a ``vireo_amd.DeviceCounts`` as AD.
        n_inits, nproc are accepted and ignored, as in the reference.
        Continues from the object's current state and appends to ``ELBO_``.
        """
        counts = device_counts(AD, DP)
        ELBO = self._fit_VB(counts, None, max_iter, min_iter, epsilon_conv,
                            delay_fit_theta, verbose)
        ELBO = ELBO + counts.binom_const()         # float32 scalar, vireo_model.py:313
        self.ELBO_ = np.append(self.ELBO_, ELBO)
