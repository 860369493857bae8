"""Development aid: A/B of one environment toggle on one bench workload -- factor + solve in two fresh processes (toggle unset / set),
device times of the best of `reps` factorisations, per-kernel-kind times, and whether the two solutions are bitwise equal.
usage: tools/ab_compare.py ENVVAR[=value] [workload] [reps]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import ipopt_amd, bench
    wl, reps, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
    n, r, c, v, neg = bench.make_workload(wl)
    s = ipopt_amd.KKTSolver(device=0)
    s.initialize_structure(n, r, c, vals=v)
    dv = torch.tensor(v, dtype=torch.float64, device="cuda"); db = torch.ones(n, dtype=torch.float64, device="cuda"); dx = torch.empty_like(db)
    tf, ts = [], []
    for i in range(reps):
        st = s.factor_device(dv.data_ptr()); tf.append(s.info().time_factor_ms)
        s.solve_device2(db.data_ptr(), dx.data_ptr()); ts.append(s.info().time_solve_ms)
    prof = s.profile(3)
    I = s.info()
    np.save(out, dx.cpu().numpy())
    print(json.dumps(dict(factor_ms=min(tf), solve_ms=min(ts), num_neg=I.num_neg, neg_ok=I.num_neg == neg, num_small=I.num_small, edits=I.num_restructures,
                          by_kind={k: round(ms / 3, 3) for k, (ms, ln) in prof.items() if ln})))
    sys.exit(0)
var = sys.argv[1]; wl = sys.argv[2] if len(sys.argv) > 2 else "synth_1e6"; reps = sys.argv[3] if len(sys.argv) > 3 else "5"
name, _, val = var.partition("=")
res = []
for on in (False, True):
    env = dict(os.environ)
    if on: env[name] = val or "1"
    out = f"/tmp/ab_{int(on)}.npy"
    p = subprocess.run([sys.executable, __file__, "--child", wl, reps, out], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line: print(p.stdout[-2000:], p.stderr[-2000:]); sys.exit(1)
    res.append(json.loads(line[-1]))
    print(("with " if on else "without ") + var, json.dumps(res[-1]), flush=True)
import numpy as np
a, b = np.load("/tmp/ab_0.npy"), np.load("/tmp/ab_1.npy")
print("bitwise equal:", bool(np.array_equal(a, b)), " max |diff|:", float(np.abs(a - b).max()))
