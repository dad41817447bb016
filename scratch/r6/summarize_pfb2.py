#!/usr/bin/env python3
"""gpurun_out/prof_r6pfb2/ (scratch/r6/prof_pfb2.sh) -> profiles/r6_pfb2_{kernel_stats,kernel_stats_serial,pmc}.csv, r6_pfb2_traffic.json, r6_pfb2_bench.json"""
import csv, json, os, collections
src, dst = os.path.join("gpurun_out", "prof_r6pfb2"), "profiles"
ours = ("mcrx::",)
for name, outn in (("pipe", "r6_pfb2_kernel_stats.csv"), ("serial", "r6_pfb2_kernel_stats_serial.csv")):
    rows = list(csv.DictReader(open(os.path.join(src, name + "_kernel_stats.csv"))))
    with open(os.path.join(dst, outn), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader()
        for r in rows:
            if any(o in r["Name"] for o in ours): w.writerow(r)
acc = collections.defaultdict(lambda: [0.0, 0])
for fn in sorted(os.listdir(src)):
    if not fn.endswith("_counter_collection.csv"): continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(os.path.join(src, fn))):
        if not any(o in r["Kernel_Name"] for o in ours): continue
        per[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, cn, _), v in per.items():
        a = acc[(k, cn)]; a[0] += v; a[1] += 1
with open(os.path.join(dst, "r6_pfb2_pmc.csv"), "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (k, c), (s, n) in sorted(acc.items()): f.write('"%s",%s,%d,%.1f\n' % (k, c, n, s / n))
ch = [k for (k, c) in acc if "channelizer_kernel" in k and c == "FETCH_SIZE"][0]
fetch, write = acc[(ch, "FETCH_SIZE")], acc[(ch, "WRITE_SIZE")]
bench = json.loads(open(os.path.join(src, "pipe.json")).read().strip().split("\n")[-1])["512ch_pfb2_front_end"]
per_launch = bench["samples_per_step"] / 2
hbm = (2.0 * fetch[0] / fetch[1] + write[0] / write[1]) * 1024.0
ser = [r for r in csv.DictReader(open(os.path.join(src, "serial_kernel_stats.csv"))) if "channelizer_kernel" in r["Name"]][0]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `LEG_CFG=serial=1 python scratch/r6/leg.py 512ch_pfb2_front_end` (scratch/r6/prof_pfb2.sh)",
           "correction": "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts wide coalesced reads at half size)",
           "workload": bench["workload"], "channels": 512, "kernel": ch, "samples_per_launch": per_launch,
           "FETCH_SIZE_KiB": fetch[0] / fetch[1], "WRITE_SIZE_KiB": write[0] / write[1], "hbm_bytes_per_launch": hbm,
           "algorithmic_bytes_per_launch": 12.0 * per_launch, "traffic_over_algorithmic": hbm / (12.0 * per_launch),
           "rocprofv3_serial_average_ns": float(ser["AverageNs"]), "achieved_GBps_on_12B_per_sample": 12.0 * per_launch / float(ser["AverageNs"]),
           "frac_of_8TBps": 12.0 * per_launch / float(ser["AverageNs"]) / 8000.0}, open(os.path.join(dst, "r6_pfb2_traffic.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(dst, "r6_pfb2_bench.json"), "w"))
print(open(os.path.join(dst, "r6_pfb2_traffic.json")).read())
