import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
import ipopt_amd
from tests.support import mirror
d = np.load('.dev_pivstat/mb2_last.npz'); mk = np.load('.dev_pivstat/mb2_marks.npz')
n, r, c = int(d['n']), d['r'], d['c']
v = d['vs'][2]; b = d['rhs'][2]
s = ipopt_amd.KKTSolver(scaling=0, delay_rounds=0)
s.initialize_structure(n, r, c, vals=v)
moved = s.delay_columns(mk['marks'])
s.values()[:] = v
x = b.copy()
st = s.multi_solve(True, x, True, 10000)
dinv = np.zeros(n); doff = np.zeros(n); pt = np.zeros(n, dtype=np.int32); lp = np.zeros(n, dtype=np.int32)
s.lib.mi355x_kkt_debug_pivots.argtypes = [C.c_void_p] * 5
assert s.lib.mi355x_kkt_debug_pivots(s._h, dinv.ctypes.data, doff.ctypes.data, pt.ctypes.data, lp.ctypes.data) == 0
sym = mirror.fetch(s)
xs, spec = mirror.factor_solve_pivoted(sym, v, b, u=1e-8, u2=1e-4, debug=True)
nbad = 0
for (sn, c0, k, m, P, ptype, dd) in spec['dbg']:
    g_d = dinv[c0:c0 + k]; g_p = lp[c0:c0 + k]; g_t = pt[c0:c0 + k]
    ok = np.array_equal(g_p, P) and np.array_equal(g_t, ptype) and np.allclose(g_d, dd, rtol=1e-6, atol=0)
    if not ok:
        nbad += 1
        if nbad <= 6:
            print('front', sn, 'c0', c0, 'k', k, 'm', m, 'level', sym['level'][sn], 'parent', sym['parent'][sn])
            print('   perm gpu', g_p[:20], 'spec', P[:20]); print('   ptype gpu', g_t[:20], 'spec', ptype[:20]); print('   dinv gpu', g_d[:8], 'spec', dd[:8])
print('fronts differing', nbad, 'of', len(spec['dbg']), 'relerr', np.abs(x - xs).max() / np.abs(xs).max())
