import ipopt_amd, numpy as np, time, sys
from tests.support import kktgen, mirror
from oracle import kkt_oracle as ko
cases = (("grid 1e-9", lambda: kktgen.hostile_grid_kkt(16,16,seed=3,tiny=1e-9)),("grid 1e-6", lambda: kktgen.hostile_grid_kkt(16,16,seed=3)),("band 1e-6", lambda: kktgen.hostile_band_kkt(2000, frac=0.15, tiny=1e-6, seed=4)),("grid 1e-9 s3 f.6", lambda: kktgen.hostile_grid_kkt(16,16,seed=3,tiny=1e-9,frac=0.6)))
for name, gen in cases:
    n,r,c,v = gen()
    K=kktgen.to_scipy(n,r,c,v); b=K@np.ones(n)
    for u in (1e-8, 1e-2):
        xo, oneg, ozero, _ = ko.factor_solve(n, r, c, v, b, u=u)
        for rounds in (30,):
            s = ipopt_amd.KKTSolver(scaling=0)
            s.initialize_structure(n,r,c,vals=v)
            nl0 = s.info().nnz_l
            t=time.time()
            x, st, edits, moved = mirror.factor_solve_delayed(s, v, b, u=u, u2=max(u,1e-4), rounds=rounds)
            res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
            print(f"{name} u={u:g} rounds={rounds}: neg={st['num_neg']} (oracle {oneg}) zero={st['num_zero']} (oracle {ozero}) forced={st['num_delay']} marks={len(st['marks'])} two={st['num_two']} edits={edits} moved={moved} nnzL {nl0}->{s.info().nnz_l} flops {s.info().flops_factor:.3g} err={np.abs(x-1).max():.2e} res={res:.1e} oerr={np.abs(xo-1).max():.2e} {time.time()-t:.1f}s", flush=True)
