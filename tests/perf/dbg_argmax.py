import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from vireo_amd import _lib, synth
from vireo_amd.counts import DeviceCounts
from vireo_amd.engine import DeviceModel
from vireo_amd.vireo_model import Vireo
from oracle import vireo_oracle as O
N,M,K,d = synth.CONFIGS[sys.argv[1] if len(sys.argv)>1 else "c3"]
w = synth.donor_workload(N,M,K,d,seed=0)
counts = DeviceCounts.from_merged(w["shape"], w["colptr"], w["rowidx"], w["ad"], w["dp"])
np.random.seed(1); host = Vireo(n_var=N,n_cell=M,n_donor=K)
dm = DeviceModel(counts,_lib.KIND_VIREO,K)
dm.set_state(host.ID_prob,host.GT_prob,host.beta_mu,host.beta_sum)
dm.set_prior(host.ID_prior,host.GT_prior,host.theta_s1_prior,host.theta_s2_prior)
tr,_ = dm.run_iters(1,0)
ID,GT,mu,sm = dm.get_state()
L = dm.get_loglik()
AD,DP = synth.as_scipy(w)
np.random.seed(1); st = O.vireo_new(M,N,K)
O.vireo_theta_step(st,AD,DP); O.vireo_gt_step(st,AD,DP); Lr = O.vireo_id_step(st,AD,DP)
bad = np.flatnonzero(ID.argmax(1)!=st.ID_prob.argmax(1))
print("n bad", bad.size, bad[:10])
print("GT max abs diff", np.abs(GT-st.GT_prob).max(), "L max abs", np.abs(L-Lr).max(), "L rel", (np.abs(L-Lr)/np.abs(Lr)).max())
print("mu", mu, st.beta_mu, "sum", sm, st.beta_sum)
for c in bad[:5]:
    print(c, "nnz", w["colptr"][c+1]-w["colptr"][c]); print(ID[c]); print(st.ID_prob[c]); print(L[c]-Lr[c])
rel = np.abs(ID-st.ID_prob)/np.maximum(st.ID_prob,1e-300)
i = np.unravel_index(rel.argmax(), rel.shape); print("worst", i, ID[i], st.ID_prob[i], rel[i])
srt = np.sort(st.ID_prob,1); gap=(srt[:,-1]-srt[:,-2])/srt[:,-1]; print("min top2 gap", gap.min(), np.sort(gap)[:5])
