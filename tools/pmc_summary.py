"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB).
FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for gfx950; tools/fetch_calib.hip measured the
counter at 0.500 of the bytes read for 4-, 8- and 16-byte lanes and for a gather (profiles/r03_fetch_size_calibration.txt).  usage: pmc_summary.py fetch.csv write.csv [out.json]
(out.json: {kernel symbol: corrected read + write bytes per launch}, read by bench.py for roofline.traffic)"""
import csv, collections, json, sys


def load(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gmrf::", "")
        a = agg[k]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | avg us | FETCH_SIZE KiB/launch | fetch x2 MB/launch | WRITE_SIZE MB/launch | x2-corrected read+write GB/s |")
print("|---|---|---|---|---|---|---|")
for k, a in sorted(f.items(), key=lambda kv: -kv[1][2]):
    n, kib, us = a
    wk = w.get(k, [1, 0.0, 0.0])
    rd = 2 * kib / n * 1024 / 1e6; wr = wk[1] / max(wk[0], 1) * 1024 / 1e6
    print(f"| {k} | {n} | {us/n:.2f} | {kib/n:.1f} | {rd:.3f} | {wr:.3f} | {(rd+wr)/(us/n)*1e3:.0f} |")

if len(sys.argv) > 3:
    table = {}
    for k, a in f.items():
        wk = w.get(k, [1, 0.0, 0.0])
        table[k] = {"launches": a[0], "read_bytes_per_launch": 2 * a[1] / a[0] * 1024,
                    "write_bytes_per_launch": wk[1] / max(wk[0], 1) * 1024, "avg_us": a[2] / a[0]}
    import os, subprocess
    try:
        head = subprocess.run(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short=12", "HEAD"],
                              capture_output=True, text=True, timeout=5).stdout.strip() or os.environ.get("GMRF_HEAD", "?")
    except Exception:
        head = os.environ.get("GMRF_HEAD", "?")
    if head == "?":
        try:
            head = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_head")).read().strip() or "?"
        except OSError:
            pass
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section; "
                         "calibrated: tools/fetch_calib.hip); bench.py --steps 1 --warmup 1 --streams 1 --no-single-problem (default batch)",
               "batch": int(os.environ.get("GMRF_PROFILE_BATCH", "64")), "head": head, "kernels": table}, open(sys.argv[3], "w"), indent=1)
