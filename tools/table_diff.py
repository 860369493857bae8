"""Where does an end-to-end run leave the reference's golden iteration table?  python tools/table_diff.py <problem> <N> <golden name> [solver ...]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prob, N, gname = sys.argv[1:4]
gold = open(os.path.join(ROOT, "tests", "golden", gname + ".iters")).read().splitlines()
for solver in (sys.argv[4:] or ["mi355x", "mi355x-pd"]):
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ipopt_mi355x_driver"), prob, N, "--solver", solver], capture_output=True, text=True, cwd="/tmp",
                         env=dict(os.environ, MKL_NUM_THREADS="1")).stdout
    it = []
    for ln in out.splitlines():
        f = ln.split()
        if len(f) >= 10 and f[0].rstrip("r").isdigit() and ln.startswith(" "):
            it.append(" ".join([f[0], f[1], f[2], f[3], f[4], f[6], f[9]]))
    first = next((i for i, (a, b) in enumerate(zip(it, gold)) if a.split()[0] != b.split()[0] or a.split()[4:] != b.split()[4:] or abs(float(a.split()[1]) - float(b.split()[1])) > 1e-7 * max(1, abs(float(b.split()[1])))), None)
    print(f"{prob} {N} {solver}: {len(it) - 1} iterations (reference {len(gold) - 1}); first line that differs: {first}")
    if first is not None:
        for i in range(max(0, first - 1), min(first + 2, len(it), len(gold))): print(f"    ours {it[i]}\n    ref  {gold[i]}")
