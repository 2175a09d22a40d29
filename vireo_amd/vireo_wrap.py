"""``vireo_wrap``: multiple random restarts, model selection, optional extra-donor search /
genotype-prior alignment, doublets -- drop-in for vireoSNP/utils/vireo_wrap.py:22-183.

Every fit runs on the GPU.  ``nproc`` is accepted for compatibility (the reference's
multiprocessing.Pool, vireo_wrap.py:74-83) and ignored: restarts are instead sharded
across GPUs when a communicator is passed (``comm=``, see vireo_amd/dist.py) -- restart i
on rank i % world, one RCCL all-gather of the ELBOs to pick the winner.

Layout of this module: ``_Plan`` resolves the argument combinations, ``_search`` runs the
restarts (vireo_amd/restarts.py) and returns the winning model, ``_keep_n_donors`` /
``_match_prior_subset`` / ``_match_prior_superset`` are the three optional re-fits, and
``_report`` / ``_result`` produce the reference's prints and its result dict.
"""
import queue
import sys
import threading
import time

import numpy as np

from .counts import device_counts
from .dist import LocalComm, gather_restart_elbos, my_restarts
from . import restarts as restarts_mod
from .restarts import DeviceRestarts, LegacyStream, _phase
from .vireo_base import donor_select, normalize, optimal_match
from .vireo_doublet import predict_doublet
from .vireo_model import Vireo

_INIT_KEYS = ("ID_prob_init", "GT_prob_init", "beta_mu_init", "beta_sum_init")
LAST_SEARCH = {}     # what the last restart search did on this rank (read by bench.py)


class _Plan:
    """What the argument combination asks for (vireo_wrap.py:36-62)."""

    def __init__(self, GT_prior, n_donor, learn_GT, n_init, n_extra_donor):
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
        self.n_donor, self.n_init, self.n_extra = n_donor, n_init, n_extra_donor
        self.learn_GT = learn_GT
        # donors searched for, and the genotype prior the search starts from
        self.search_donors = int(n_donor + n_extra_donor)
        self.search_prior = None
        if GT_prior is not None and self.search_donors <= GT_prior.shape[1]:
            self.search_prior = GT_prior.copy()
            self.search_donors = GT_prior.shape[1]


def _template(counts, n_donor, learn_GT, GT_prior, kwargs, **state):
    """A host Vireo with the job's shapes, flags and priors that consumes no random numbers:
    whatever the caller does not pin gets a placeholder that a device state replaces."""
    n_var, n_cell = counts.shape
    n_GT = kwargs.get("n_GT", 3)
    rows = n_var if kwargs.get("ASE_mode", False) else 1
    # placeholders: ONE row each through the constructor (it normalises what it is given: 45 MB
    # of ones at c3, 27 ms per vireo_wrap), then untouched arrays of the full shape -- a fitted
    # device state or a broadcast replaces them before anything reads them
    init = dict(ID_prob_init=np.ones((1, n_donor)),
                GT_prob_init=np.ones((1, n_donor, n_GT)) if GT_prior is None else GT_prior)
    init.update({k: v for k, v in kwargs.items() if k in _INIT_KEYS and v is not None})
    init.update(state)
    flags = {k: v for k, v in kwargs.items() if k not in _INIT_KEYS}
    m = Vireo(n_var=n_var, n_cell=n_cell, n_donor=n_donor, learn_GT=learn_GT, **init, **flags)
    if m.ID_prob.shape[0] != n_cell:
        m.ID_prob = np.empty((n_cell, n_donor))
    if m.GT_prob.shape[0] != n_var:
        m.GT_prob = np.empty((n_var, n_donor, n_GT))
    m.set_prior(GT_prior=GT_prior)
    assert m.beta_mu.shape[0] in (1, rows)
    return m


def _bcast_model(comm, model, root, dm=None, force=False):
    """every rank leaves with the root's fitted state (the reference continues with
    ``_models_all[_idx]``, vireo_wrap.py:90-94).

    ``dm``: this rank's device model -- on the root it holds the winner's final state.  Under RCCL
    the four state arrays then travel DEVICE TO DEVICE (``comm.bcast_model``: ncclBroadcast from /
    into the models' HBM buffers, one group call) and the other ranks download them once for their
    result dict; the host route (the root's arrays staged back up, broadcast, downloaded: twice
    across the host on every rank) stays for communicators without a device path (the one-device
    TcpComm harness, the CPU tests' gloo).  ``force``: also at world 1 (tests, bench.py's timing)."""
    if comm.world == 1 and not force:
        return
    if dm is not None and hasattr(comm, "bcast_model"):
        comm.bcast_model(dm, root)
        if comm.rank != root:
            model._pull(dm, want_GT=True)
    else:
        for name in ("ID_prob", "GT_prob", "beta_mu", "beta_sum"):
            setattr(model, name, comm.bcast(getattr(model, name), root))
    n = comm.bcast(np.array([float(len(model.ELBO_))]), root)
    trace = model.ELBO_ if comm.rank == root else np.zeros(int(n[0]))
    model.ELBO_ = comm.bcast(trace, root)


def _one_ahead(gen, depth=1):
    """Iterate ``gen`` with its next ``depth`` items being produced on a helper thread while the
    caller works on the current one (the library calls on both sides release the GIL).  The helper has
    finished -- or failed, and the error is re-raised here -- when the iteration ends."""
    box = queue.Queue(maxsize=max(1, depth))
    stop = threading.Event()
    done = object()

    def hand_over(item):
        """False when the consumer has gone away"""
        while not stop.is_set():
            try:
                box.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def produce():
        try:
            for item in gen:
                if not hand_over(item):
                    gen.close()         # (its `finally` runs here, on this thread, not at collection)
                    return
            hand_over(done)
        except BaseException as e:      # noqa: BLE001 -- handed to the consumer
            hand_over(e)

    th = threading.Thread(target=produce, name="vireo-restart-draws", daemon=True)
    th.start()
    try:
        while True:
            item = box.get()
            if item is done:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        th.join()


def _search(counts, plan, comm, max_iter_init, delay_fit_theta, kwargs, restarts_cls):
    """The n_init restarts (vireo_wrap.py:64-94): every rank walks the random stream of all of
    them, fits its own share, the ELBOs are all-gathered, the first maximum wins and its owner
    refines it (unless extra donors are still to be dropped) and broadcasts the state."""
    n_var, n_cell = counts.shape
    K, T = plan.search_donors, kwargs.get("n_GT", 3)
    with _phase("template"):
        tmpl = _template(counts, K, plan.learn_GT, plan.search_prior, kwargs)
    # what one reference constructor draws, in its order (vireo_model.py:98,103)
    fixed_ID = kwargs.get("ID_prob_init")
    ID0 = None if fixed_ID is None else normalize(fixed_ID, axis=1)
    # a genotype prior is every restart's initial GT_prob; the first constructor sees it before
    # set_prior clips it in place (vireo_model.py:132-133), the later ones after
    GT0 = GT0_first = None
    if plan.search_prior is not None:
        GT0_first, GT0 = tmpl.GT_prob.copy(), normalize(plan.search_prior)
    stream = LegacyStream()
    mine = set(my_restarts(plan.n_init, comm.rank, comm.world))
    with _phase("device_models"):
        if hasattr(restarts_cls, "submit"):       # packs restarts into one device model
            runner = restarts_cls(counts, tmpl, n_owned=len(mine))
            batch = runner.batch
        else:
            runner, batch = restarts_cls(counts, tmpl), 1

    stage = runner.stage if batch == 1 and getattr(runner, "can_stage", False) else None

    def draws():
        """this rank's restarts in order, each with what its constructor draws; the draws of
        the other ranks' restarts are consumed without being formed -- runs of them in ONE skip
        (far skips are a jump of the generator: vrx_mt19937_random_sample, cached per length)"""
        per_restart = (n_cell * K if ID0 is None else 0) + (n_var * K * T if GT0 is None else 0)
        consumed = 0          # restarts whose draws the global stream has passed
        try:
            for im in range(plan.n_init):
                if im in mine:
                    if im > consumed:
                        stream.skip((im - consumed) * per_restart)
                    consumed = im + 1
                    ID_raw = stream.rand(n_cell, K) if ID0 is None else None
                    GT_raw = stream.rand(n_var, K, T) if GT0 is None else None
                    if stage is not None and ID_raw is not None and GT_raw is not None:
                        # (this generator runs on the helper thread: the upload overlaps the fit
                        #  of the restart before, restarts.DeviceRestarts.stage)
                        ID_raw, GT_raw = stage(ID_raw, GT_raw), None
                    yield (im, ID_raw, GT_raw)
        finally:
            # the global stream ends where the reference's would (all n_init constructors run
            # before the first fit, vireo_wrap.py:66-71) -- also when the consumer stops early
            if consumed < plan.n_init:
                stream.skip((plan.n_init - consumed) * per_restart)

    local = {}
    t_search = time.perf_counter()
    # the generator runs one restart ahead on a helper thread: the Mersenne Twister (host, serial)
    # and the fit (device) overlap; the helper is the only user of the stream meanwhile
    try:
        try:
            for im, ID_raw, GT_raw in _one_ahead(draws(), depth=batch):
                args = (im, ID_raw, GT_raw, ID0, GT0_first if im == 0 else GT0, max_iter_init,
                        delay_fit_theta)
                if batch > 1:
                    runner.submit(*args)         # fitted `batch` at a time
                else:
                    local[im] = runner.run(*args)
        except BaseException:
            if hasattr(runner, "cancel"):        # (the helper thread may be waiting for a staging buffer)
                runner.cancel()
            raise
        if batch > 1:
            local.update(runner.flush())
        if restarts_mod.PHASES is not None:
            restarts_mod.PHASES["search_wall"] = time.perf_counter() - t_search
        with _phase("gather"):
            elbo_all = gather_restart_elbos(comm, plan.n_init, local)
        best = int(np.argmax(elbo_all))              # first max wins, vireo_wrap.py:90-91
        owner = best % comm.world
        model = runner.winner(best, refine=plan.n_extra == 0) if comm.rank == owner else tmpl
        LAST_SEARCH.clear()
        LAST_SEARCH.update(restarts=len(local), restart_iterations=getattr(runner, "iterations", 0),
                           final_iterations=getattr(runner, "final_iterations", 0), best=best,
                           owner=owner, batch=batch)
        with _phase("broadcast"):     # (before the runner goes: its device model is the broadcast's buffer)
            _bcast_model(comm, model, owner, dm=getattr(runner, "dm", None))
    finally:
        runner.close()       # device models and staging buffers go now, not at garbage collection
    return model, elbo_all


def _keep_n_donors(counts, plan, found, extra_donor_mode, delay_fit_theta, kwargs):
    """More donors were searched than asked for: keep n_donor of them (by size or by genotype
    distance) and fit again from their assignments (vireo_wrap.py:95-105)."""
    n_var, n_cell = counts.shape
    start = dict(kwargs, beta_mu_init=found.beta_mu, beta_sum_init=found.beta_sum,
                 ID_prob_init=donor_select(found.GT_prob, found.ID_prob, plan.n_donor,
                                           mode=extra_donor_mode))
    model = Vireo(n_var=n_var, n_cell=n_cell, n_donor=plan.n_donor, learn_GT=plan.learn_GT,
                  GT_prob_init=plan.search_prior, **start)   # draws GT_prob when there is no prior
    model.set_prior(GT_prior=plan.search_prior)
    model.fit(counts, None, min_iter=5, delay_fit_theta=delay_fit_theta, verbose=False)
    return model


def _match_prior_subset(counts, plan, found, GT_prior, kwargs):
    """The prior knows more donors than are in the pool: keep the n_donor largest found ones
    and fit with their genotypes fixed (vireo_wrap.py:111-119)."""
    by_size = np.argsort(np.sum(found.ID_prob, axis=0))[::-1]
    kept = GT_prior[:, by_size[:plan.n_donor], :]
    n_var, n_cell = counts.shape
    model = Vireo(n_var=n_var, n_cell=n_cell, n_donor=plan.n_donor, learn_GT=False,
                  GT_prob_init=kept, **kwargs)
    model.fit(counts, None, min_iter=20, verbose=False)
    return model


def _match_prior_superset(counts, plan, found, GT_prior, kwargs):
    """The pool holds more donors than the prior knows: align the known ones to the found
    genotypes, put them first, and fit again with the learned genotypes of the others as their
    prior (vireo_wrap.py:121-136)."""
    known = optimal_match(GT_prior, found.GT_prob)[1]
    order = np.append(known, np.delete(np.arange(plan.n_donor), known))
    mixed = found.GT_prob.copy()
    mixed[:, known, :] = GT_prior
    mixed = mixed[:, order, :]
    model = _template(counts, plan.n_donor, plan.learn_GT, mixed, kwargs,
                      ID_prob_init=found.ID_prob[:, order], beta_mu_init=found.beta_mu,
                      beta_sum_init=found.beta_sum)
    model.fit(counts, None, min_iter=20, verbose=False)
    return model


def _report(model):
    print("[vireo] allelic rate mean and concentrations:")
    print(np.round(model.beta_mu, 3))
    print(np.round(model.beta_sum, 1))
    print("[vireo] donor size before removing doublets:")
    sizes = np.sum(model.ID_prob, axis=0)
    print("\t".join("donor%d" % k for k in range(len(sizes))))
    print("\t".join("%.0f" % s for s in sizes))


def _result(model, ID_prob, doublet_prob, doublet_LLR, elbo_all):
    """the reference's result dict (vireo_wrap.py:170-183)"""
    s1, s2 = model.beta_mu * model.beta_sum, (1 - model.beta_mu) * model.beta_sum
    return dict(ID_prob=ID_prob, GT_prob=model.GT_prob, doublet_LLR=doublet_LLR,
                doublet_prob=doublet_prob, theta_shapes=np.append(s1, s2, axis=0),
                theta_mean=model.beta_mu, theta_sum=model.beta_sum, ambient_Psi=None,
                Psi_var=None, Psi_LLRatio=None, LB_list=elbo_all, LB_doublet=model.ELBO_[-1])


def vireo_wrap(AD, DP, GT_prior=None, n_donor=None, learn_GT=True, n_init=20,
               random_seed=None, check_doublet=True, max_iter_init=20, delay_fit_theta=3,
               n_extra_donor=0, extra_donor_mode="distance",
               check_ambient=False, nproc=4, comm=None, **kwargs):
    """Run vireo with multiple initialisations; returns the reference's result dict
    (keys: vireo_wrap.py:170-183)."""
    comm = LocalComm() if comm is None else comm
    plan = _Plan(GT_prior, n_donor, learn_GT, n_init, n_extra_donor)
    # every restart and the final fit run on ONE device problem (vireo_wrap.py:64-94): tell the builder how
    # many iterations that is, so that it can spend a one-off effort that pays back over them
    counts = device_counts(AD, DP, expected_iterations=-(-plan.n_init // comm.world) * max_iter_init + 200)
    if check_ambient:
        raise NotImplementedError("check_ambient (experimental in the reference, "
                                  "vireo.py:79-81) is out of scope of vireo_amd")
    if random_seed is not None:                       # the ONLY seeding, vireo_wrap.py:53-54
        np.random.seed(random_seed)

    model, elbo_all = _search(counts, plan, comm, max_iter_init, delay_fit_theta, kwargs,
                              DeviceRestarts)
    if plan.n_extra != 0:
        model = _keep_n_donors(counts, plan, model, extra_donor_mode, delay_fit_theta, kwargs)
    print("[vireo] lower bound ranges [%.1f, %.1f, %.1f]"
          % (np.min(elbo_all), np.median(elbo_all), np.max(elbo_all)))

    if GT_prior is not None and plan.n_donor < GT_prior.shape[1]:
        model = _match_prior_subset(counts, plan, model, GT_prior, kwargs)
    elif GT_prior is not None and plan.n_donor > GT_prior.shape[1]:
        model = _match_prior_superset(counts, plan, model, GT_prior, kwargs)
    _report(model)

    if check_doublet:                                 # vireo_wrap.py:151-156
        doublet_prob, ID_prob, doublet_LLR = predict_doublet(model, counts, None)
    else:
        n_cell, K = counts.shape[1], plan.n_donor
        ID_prob, doublet_LLR = model.ID_prob, np.zeros(n_cell)
        doublet_prob = np.zeros((n_cell, int(K * (K - 1) / 2)))
    return _result(model, ID_prob, doublet_prob, doublet_LLR, elbo_all)
