cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -o mp -- python /root/repo/tools/match_time.py lukvle1_1e6 > /tmp/mp.log 2>&1
f=$(find /tmp/mp -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-150
t=$(find /tmp/mp -name "*kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last matching computation: from the last k_match_init to the following k_match_final
idx = [i for i, r in enumerate(rows) if "k_match_init" in r["Kernel_Name"]]
i0 = idx[-1]; i1 = [i for i, r in enumerate(rows) if i > i0 and "k_match_final" in r["Kernel_Name"]][0]
t0 = int(rows[i0]["Start_Timestamp"]); busy = 0
for r in rows[i0:i1 + 1]:
    busy += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("launches", i1 - i0 + 1, "wall us", (int(rows[i1]["End_Timestamp"]) - t0) / 1e3, "busy us", busy / 1e3)
gaps = []
for a, b in zip(rows[i0:i1], rows[i0 + 1:i1 + 1]):
    gaps.append((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3)
gaps.sort(reverse=True)
print("largest gaps us", [round(g) for g in gaps[:16]])
for r in rows[i0:i0 + 40]:
    print(round((int(r["Start_Timestamp"]) - t0) / 1e3, 1), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1), r["Kernel_Name"].split("(")[0][-20:], r.get("Grid_Size_X", r.get("Grid_Size")))
PY
