"""``vireo_wrap``: multiple random restarts, model selection, optional extra-donor search /
genotype-prior alignment, doublets -- drop-in for vireoSNP/utils/vireo_wrap.py:22-183.

Every fit runs on the GPU.  ``nproc`` is accepted for compatibility (the reference's
multiprocessing.Pool, vireo_wrap.py:74-83) and ignored: restarts are instead sharded
across GPUs when a communicator is passed (``comm=``, see vireo_amd/dist.py) -- restart i
on rank i % world, one RCCL all-gather of the ELBOs to pick the winner.
"""
import sys

import numpy as np

from .counts import device_counts
from .dist import LocalComm, gather_restart_elbos, my_restarts
from .vireo_base import donor_select, optimal_match
from .vireo_doublet import predict_doublet
from .vireo_model import Vireo


def _bcast_model(comm, model, root):
    """every rank leaves with the root's fitted state."""
    if comm.world == 1:
        return
    model.ID_prob = comm.bcast(model.ID_prob, root)
    model.GT_prob = comm.bcast(model.GT_prob, root)
    model.beta_mu = comm.bcast(model.beta_mu, root)
    model.beta_sum = comm.bcast(model.beta_sum, root)
    n = comm.bcast(np.array([float(len(model.ELBO_))]), root)
    trace = model.ELBO_ if comm.rank == root else np.zeros(int(n[0]))
    model.ELBO_ = comm.bcast(trace, root)


def _shell_like(n_var, n_cell, n_donor, learn_GT, GT_prior, kwargs):
    """an un-initialised Vireo with the right shapes and priors; consumes no random numbers"""
    n_GT = kwargs.get("n_GT", 3)
    ase = kwargs.get("ASE_mode", False)
    rows = n_var if ase else 1
    m = Vireo(n_var=n_var, n_cell=n_cell, n_donor=n_donor, learn_GT=learn_GT,
              ID_prob_init=np.ones((n_cell, n_donor)),
              GT_prob_init=np.ones((n_var, n_donor, n_GT)),
              beta_mu_init=np.zeros((rows, n_GT)), beta_sum_init=np.zeros((rows, n_GT)),
              **{k: v for k, v in kwargs.items()
                 if k not in ("ID_prob_init", "GT_prob_init", "beta_mu_init", "beta_sum_init")})
    m.set_prior(GT_prior=GT_prior)
    return m


def vireo_wrap(AD, DP, GT_prior=None, n_donor=None, learn_GT=True, n_init=20,
               random_seed=None, check_doublet=True, max_iter_init=20, delay_fit_theta=3,
               n_extra_donor=0, extra_donor_mode="distance",
               check_ambient=False, nproc=4, comm=None, **kwargs):
    """Run vireo with multiple initialisations; returns the reference's result dict
    (keys: vireo_wrap.py:170-183)."""
    if comm is None:
        comm = LocalComm()
    counts = device_counts(AD, DP)
    n_var, n_cell = counts.shape

    if learn_GT == False and n_extra_donor > 0:      # noqa: E712
        print("Searching from extra donors only works with learn_GT")
        n_extra_donor = 0
    if n_donor is None:
        if GT_prior is None:
            print("[vireo] Error: requiring n_donor or GT_prior.")
            sys.exit()
        n_donor = GT_prior.shape[1]
    if learn_GT is False and n_init > 1:
        print("GT is fixed, so use a single initialization")
        n_init = 1
    if check_ambient:
        raise NotImplementedError("check_ambient (experimental in the reference, "
                                  "vireo.py:79-81) is out of scope of vireo_amd")

    if random_seed is not None:                       # the ONLY seeding, vireo_wrap.py:53-54
        np.random.seed(random_seed)

    GT_prior_use = None
    n_donor_use = int(n_donor + n_extra_donor)
    if GT_prior is not None and n_donor_use == GT_prior.shape[1]:
        GT_prior_use = GT_prior.copy()
    elif GT_prior is not None and n_donor_use < GT_prior.shape[1]:
        GT_prior_use = GT_prior.copy()
        n_donor_use = GT_prior.shape[1]

    # Every rank draws every restart's initial state in the reference's order so the RNG
    # stream is consumed identically (vireo_wrap.py:66-71), but keeps only its own share.
    mine = set(my_restarts(n_init, comm.rank, comm.world))
    models = {}
    for im in range(n_init):
        mdl = Vireo(n_var=n_var, n_cell=n_cell, n_donor=n_donor_use, learn_GT=learn_GT,
                    GT_prob_init=GT_prior_use, **kwargs)
        mdl.set_prior(GT_prior=GT_prior_use)
        if im in mine:
            models[im] = mdl

    for im in sorted(models):                         # vireo_wrap.py:84-87
        models[im].fit(counts, None, min_iter=5, max_iter=max_iter_init,
                       delay_fit_theta=delay_fit_theta, verbose=False)

    # select the best initialisation (first max wins, vireo_wrap.py:90-91)
    elbo_all = gather_restart_elbos(comm, n_init, {i: m.ELBO_[-1] for i, m in models.items()})
    best = int(np.argmax(elbo_all))
    owner = best % comm.world
    if comm.rank == owner:
        modelCA = models[best]
        if n_extra_donor == 0:
            modelCA.fit(counts, None, min_iter=5, verbose=False)
    else:      # a shell that receives the winner's state (built without touching the RNG)
        modelCA = _shell_like(n_var, n_cell, n_donor_use, learn_GT, GT_prior_use, kwargs)
    _bcast_model(comm, modelCA, owner)
    models.clear()

    if n_extra_donor != 0:                            # vireo_wrap.py:95-105
        _ID_prob = donor_select(modelCA.GT_prob, modelCA.ID_prob, n_donor,
                                mode=extra_donor_mode)
        modelCA = Vireo(n_var=n_var, n_cell=n_cell, n_donor=n_donor, learn_GT=learn_GT,
                        GT_prob_init=GT_prior_use, ID_prob_init=_ID_prob,
                        beta_mu_init=modelCA.beta_mu, beta_sum_init=modelCA.beta_sum,
                        **kwargs)
        modelCA.set_prior(GT_prior=GT_prior_use)
        modelCA.fit(counts, None, min_iter=5, delay_fit_theta=delay_fit_theta, verbose=False)

    print("[vireo] lower bound ranges [%.1f, %.1f, %.1f]"
          % (np.min(elbo_all), np.median(elbo_all), np.max(elbo_all)))

    # run again when the genotype prior has more / fewer donors than asked for
    if GT_prior is not None and n_donor < GT_prior.shape[1]:        # vireo_wrap.py:111-119
        _donor_cnt = np.sum(modelCA.ID_prob, axis=0)
        _donor_idx = np.argsort(_donor_cnt)[::-1]
        GT_prior_use = GT_prior[:, _donor_idx[:n_donor], :]
        modelCA = Vireo(n_var=n_var, n_cell=n_cell, n_donor=n_donor, learn_GT=False,
                        GT_prob_init=GT_prior_use, **kwargs)
        modelCA.fit(counts, None, min_iter=20, verbose=False)
    elif GT_prior is not None and n_donor > GT_prior.shape[1]:      # vireo_wrap.py:121-136
        GT_prior_use = modelCA.GT_prob.copy()
        idx = optimal_match(GT_prior, GT_prior_use)[1]
        GT_prior_use[:, idx, :] = GT_prior
        _idx_order = np.append(idx, np.delete(np.arange(n_donor), idx))
        GT_prior_use = GT_prior_use[:, _idx_order, :]
        ID_prob_use = modelCA.ID_prob[:, _idx_order]
        modelCA = Vireo(n_var=n_var, n_cell=n_cell, n_donor=n_donor, learn_GT=learn_GT,
                        ID_prob_init=ID_prob_use, beta_mu_init=modelCA.beta_mu,
                        beta_sum_init=modelCA.beta_sum, GT_prob_init=GT_prior_use, **kwargs)
        modelCA.set_prior(GT_prior=GT_prior_use)
        modelCA.fit(counts, None, min_iter=20, verbose=False)

    print("[vireo] allelic rate mean and concentrations:")
    print(np.round(modelCA.beta_mu, 3))
    print(np.round(modelCA.beta_sum, 1))
    print("[vireo] donor size before removing doublets:")
    _donor_cnt = np.sum(modelCA.ID_prob, axis=0)
    print("\t".join(["donor%d" % x for x in range(len(_donor_cnt))]))
    print("\t".join(["%.0f" % x for x in _donor_cnt]))

    if check_doublet:                                 # vireo_wrap.py:151-156
        doublet_prob, ID_prob, doublet_LLR = predict_doublet(modelCA, counts, None)
    else:
        ID_prob = modelCA.ID_prob
        doublet_prob = np.zeros((n_cell, int(n_donor * (n_donor - 1) / 2)))
        doublet_LLR = np.zeros(n_cell)

    theta_shapes = np.append(modelCA.beta_mu * modelCA.beta_sum,
                             (1 - modelCA.beta_mu) * modelCA.beta_sum, axis=0)
    return {
        'ID_prob': ID_prob, 'GT_prob': modelCA.GT_prob, 'doublet_LLR': doublet_LLR,
        'doublet_prob': doublet_prob, 'theta_shapes': theta_shapes,
        'theta_mean': modelCA.beta_mu, 'theta_sum': modelCA.beta_sum,
        'ambient_Psi': None, 'Psi_var': None, 'Psi_LLRatio': None,
        'LB_list': elbo_all, 'LB_doublet': modelCA.ELBO_[-1],
    }
