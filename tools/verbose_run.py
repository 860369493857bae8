import sys; sys.path.insert(0,'/root/repo')
import ipopt_amd, bench, numpy as np
n,r,c,v,neg=bench.make_workload(sys.argv[1])
s=ipopt_amd.KKTSolver(verbose=1); s.initialize_structure(n,r,c,vals=v); s.values()[:]=v
x=np.ones(n); print(s.multi_solve(True,x,True,neg))
