"""Summarise a rocprofv3 kernel-trace CSV by (kernel, grid): per-job time, launches, average."""
import csv, collections, sys
path = sys.argv[1]; jobs = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrf::", "")
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[(name, g)][0] += 1; agg[(name, g)][1] += d
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches; per job {tot/1e3/jobs:.2f} ms")
byname = collections.defaultdict(lambda: [0, 0.0])
for (n, g), v in agg.items():
    byname[n][0] += v[0]; byname[n][1] += v[1]
for n, v in sorted(byname.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{v[1]/jobs/1e3:9.3f} ms/job  n={v[0]/jobs:8.1f}  avg {v[1]/v[0]:8.2f} us  {n}")
print("--- by grid")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{v[1]/jobs:10.1f} us/job  n={v[0]/jobs:7.1f}  avg {v[1]/v[0]:8.2f} us  {k}")
