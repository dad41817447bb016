#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (scratch/prof_rs.sh) -> profiles/<pre>_resamp_roofline.csv: per workload and kernel the rocprofv3 average
duration, the HBM bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, KiB) and the algorithmic bytes."""
import csv, collections, os, sys
tag, pre = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r2")
src = os.path.join("gpurun_out", "prof_" + tag)
n = 77_900_000 // 1024 * 1024
W = {  # workload -> {kernel substring: algorithmic bytes per launch}
    "rs0.5": {"arbitrary": n * 8 + n // 2 * 8},
    "rs0.8": {"arbitrary": n * 8 + int(n * 0.8) * 8},
    "rs0.37": {"arbitrary": n * 8 + int(n * 0.37) * 8},      # (round 6: the half-band decimator is folded into the arbitrary stage's kernel: 8 B in, 0.37 x 8 B out)
    "rs2.0": {"arbitrary": n // 4 * 8 + n // 2 * 8},
    "pfb1024": {"pfb2": 100000 * 512 * 24}, "pfb128": {"pfb2": 400000 * 64 * 24}, "pfb16": {"pfb2": 2000000 * 8 * 24},
}
out = open(os.path.join("profiles", "%s_resamp_roofline.csv" % pre), "w")
out.write("workload,kernel,launches,avg_us,algorithmic_MB,hbm_MB_pmc,achieved_GBps_algorithmic,frac_of_8TBps\n")
for w, ks in W.items():
    st = list(csv.DictReader(open(os.path.join(src, w + "_stats_kernel_stats.csv"))))
    def pmc(c):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(os.path.join(src, "%s_pmc_%s_counter_collection.csv" % (w, c)))):
            per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        acc = collections.defaultdict(list)
        for (k, _), v in per.items():
            acc[k].append(v)
        return {k: sum(v) / len(v) for k, v in acc.items()}
    f, wr = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
    for sub, alg in ks.items():
        r = [x for x in st if sub in x["Name"]][0]
        kn = r["Name"]
        hbm = (2 * f.get(kn, 0) + wr.get(kn, 0)) * 1024
        us = float(r["AverageNs"]) / 1e3
        gb = alg / us / 1e3
        out.write('%s,"%s",%s,%.1f,%.1f,%.1f,%.0f,%.3f\n' % (w, kn.split("(")[0].replace("void ", ""), r["Calls"], us, alg / 1e6, hbm / 1e6, gb, gb / 8000))
out.close()
print(open(os.path.join("profiles", "%s_resamp_roofline.csv" % pre)).read())
