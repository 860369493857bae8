"""k_front_df (the persistent data-flow kernel over runs of small-front levels) against the per-level kernels, on recorded KKT systems:
   python tools/df_check.py .dev_pivstat/lukvle5_calls.npz"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ipopt_amd
z = np.load(sys.argv[1])
n, r, c, vals, rhs, neg = int(z["n"]), z["r"], z["c"], z["vals"], z["rhs"], z["neg"]
out = {}
for mode in ("df", "nodf"):
    if mode == "nodf": os.environ["MI355X_KKT_DISABLE"] = "front_df"
    else: os.environ.pop("MI355X_KKT_DISABLE", None)
    s = ipopt_amd.KKTSolver(); s.initialize_structure(n, r, c, vals=vals[0])
    res = []
    for i in range(len(vals)):
        s.values()[:] = vals[i]
        b_ = np.array(rhs[i], dtype=np.float64).reshape(-1)[:n]; x = b_.copy()
        st = s.multi_solve(True, x, check_neg_evals=True, number_of_neg_evals=int(neg[i]))
        I = s.info()
        Ax = np.zeros(n); np.add.at(Ax, r - 1, vals[i] * x[c - 1]); off = r != c; np.add.at(Ax, c[off] - 1, vals[i][off] * x[r[off] - 1])
        res.append((int(st), s.number_of_neg_evals(), I.num_two, I.num_delayed, x.copy(), np.abs(Ax - b_).max() / max(np.abs(b_).max(), 1e-300)))
    out[mode] = res
for i in range(len(vals)):
    a, b = out["df"][i], out["nodf"][i]
    d = np.abs(a[4] - b[4]).max()
    print(i, "required", int(neg[i]), "df:", a[:4], "nodf:", b[:4], "max|dx|", d, "rel", d / np.abs(b[4]).max(), "res df %.1e nodf %.1e" % (a[5], b[5]), "bitwise" if np.array_equal(a[4], b[4]) else "DIFF")
