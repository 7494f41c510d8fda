import sys; sys.path.insert(0,"tests"); import numpy as np, mpc_setup as S
om, gm, rb = S.make_cent_pair(4, 3)
X = S.random_states(rb, 4); w=[0,0,0,0]
for i in range(8):
    om.iterate(X); gm.iterate(X)
    e=[S.rel_err(om.xs,gm.xs), S.rel_err(om.us, gm.us), S.rel_err(om.lams, gm.lams), S.rel_err(om.vs, gm.vs)]
    w=[max(a,b) for a,b in zip(w,e)]
print("closed-loop worst rel err xs us lams vs", w, "K0", S.rel_err(om.K0, gm.K0), 'alphas equal', np.array_equal(om.info[:,2], gm.info[:,2]))
