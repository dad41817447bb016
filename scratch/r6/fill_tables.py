"""Rewrite the result rows of DESIGN.md section 4.5, README.md's status table and BASELINE.md section 3 from ONE bench line:
   python scratch/r6/fill_tables.py <tag>     (profiles/r6_<tag>_bench.json, _bench_pipeline.json, _duplex.json)"""
import json, re, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
P = lambda n: os.path.join(ROOT, "profiles", "r6_%s_%s" % (tag, n))
d = json.load(open(P("bench.json"))); c = d["configs"]
pipe = json.load(open(P("bench_pipeline.json"))); dup = json.load(open(P("duplex.json")))
g = lambda x: "%.1f" % (x / 1000.0)
def sp(x):
    s = "%d" % round(x)
    return s[:-3] + " " + s[-3:] if len(s) > 3 else s
rng = lambda v: "%s (%s - %s)" % (sp(v["value"]), sp(v["value_min"]), sp(v["value_max"]))
c1, c1l, c1v = c["8ch"], c["8ch_long_pushes"], c["8ch_v27"]
m48, c2, pf, dx = c["512ch_m48_reference_app_defaults"], c["64ch_m256_qam16_resamp"], c["512ch_pfb2_front_end"], c["256ch_duplex_one_gpu"]
r = d["roofline"]; vi = r["vector_issue"]
def sub_row(text, start, new):
    """replace the one table row (a whole line) that starts with `start`"""
    lines = text.split("\n"); n = 0
    for i, l in enumerate(lines):
        if l.startswith(start): lines[i] = new; n += 1
    assert n == 1, (n, start)
    return "\n".join(lines)

# ---------------- DESIGN.md section 4.5
p = os.path.join(ROOT, "DESIGN.md"); s = open(p).read()
s = re.sub(r"profiles/r6_t\d_bench\.json`\*\*\n", "profiles/r6_%s_bench.json`**\n" % tag, s)
s = re.sub(r"`profiles/r6_t\d_\*` are rocprofv3", "`profiles/r6_%s_*` are rocprofv3" % tag, s)
s = sub_row(s, "| `value` (configs[3]", "| `value` (configs[3] on one GPU: 512 ch, M = 64) | %s (%s-%s) | %.1f %% of 16 B/sample |" % (g(d["value"]), g(d["value_min"]), g(d["value_max"]), d["value"] * 1e6 * 16 / 8e12 * 100))
s = sub_row(s, "| `value_with_harvest`", "| `value_with_harvest` | %s = %.2f × | |" % (g(d["value_with_harvest"]), d["value_with_harvest"] / d["value"]))
s = sub_row(s, "| `value_aperiodic`", "| `value_aperiodic` (ragged lengths, silences) | %s = %.2f × | |" % (g(d["value_aperiodic"]), d["value_aperiodic"] / d["value"]))
s = sub_row(s, "| `value_awgn30`", "| `value_awgn30` (30 dB AWGN on the wideband samples) | %s | |" % g(d["value_awgn30"]))
s = sub_row(s, "| `value_without_framesyms`", "| `value_without_framesyms` (`skip_framesyms = 2`) | %s | |" % g(d["value_without_framesyms"]))
s = sub_row(s, "| `value_default_hw_queues`", "| `value_default_hw_queues` (the same loop in a process without `GPU_MAX_HW_QUEUES=8`; the knob is reported in `config.runtime_knobs`) | %s | the knob is worth 0.1-1.5 %% here; it matters for the harvest legs |" % g(d["value_default_hw_queues"]))
s = sub_row(s, "| configs[1] 8 ch", "| configs[1] 8 ch / long pushes / + K = 7 | %s (%s-%s) / %s / **%s** (r5 driver: 81.7 / — / 47.7; the round's first profiled library, `r6_t1`: 80.2 / 139.1 / 46.8) | %.1f / %.1f / %.1f %% |" % (
    g(c1["value"]), g(c1["value_min"]), g(c1["value_max"]), g(c1l["value"]), g(c1v["value"]), c1["frac_of_roofline"] * 100, c1l["frac_of_roofline"] * 100, c1v["frac_of_roofline"] * 100))
s = sub_row(s, "| 512 ch at the applications' defaults", "| 512 ch at the applications' defaults M = 48 | %s | %.1f %% |" % (g(m48["value"]), m48["frac_of_roofline"] * 100))
s = sub_row(s, "| **configs[2]**", "| **configs[2]** 64 ch M = 256 QAM16 + resampler | **%s** (r5 driver: 88.7) | %.1f %% of 20 B |" % (g(c2["value"]), c2["frac_of_roofline"] * 100))
s = sub_row(s, "| **512 ch behind the oversampled front end**", "| **512 ch behind the oversampled front end** (`front_end = 1`, new, §4.7) | **%s** (stage by stage, `front_end = 2`: 58.9) | %.1f %% of 16 B; the kernel alone %.1f %% of 8 TB/s on 12 B/sample, traffic 1.07 × |" % (g(pf["value"]), pf["frac_of_roofline"] * 100, pf["roofline"]["frac"] * 100))
s = sub_row(s, "| configs[4] on one GPU", "| configs[4] on one GPU, 256 ch both ways | %s as a leg of this line; %s in a process of its own (`profiles/r6_%s_duplex.json`) | %.1f %% of 28 B |" % (g(dx["value"]), g(dup["value"]), tag, dx["frac_of_roofline"] * 100))
s = sub_row(s, "| `bench.py --pipeline`", "| `bench.py --pipeline` (the multi-GPU code path on one GPU; `profiles/r6_%s_bench_pipeline.json`) | %s = %.2f × | |" % (tag, g(pipe["value"]), pipe["value"] / d["value"]))
s = sub_row(s, "| CPU oracle, one thread / 16", "| CPU oracle, one thread / 16 | %.4f / %.3f | |" % (d["cpu_baseline"]["value"] / 1000, d["cpu_baseline"]["all_cores"]["value"] / 1000))
open(p, "w").write(s)

# ---------------- README.md
p = os.path.join(ROOT, "README.md"); s = open(p).read()
s = re.sub(r"from one file, `profiles/r6_t\d_bench\.json`", "from one file, `profiles/r6_%s_bench.json`" % tag, s)
s = sub_row(s, "| one MI355X (BASELINE.md section 3) |", "| one MI355X (BASELINE.md section 3) | 512 channels **%s Gsample/s** (184.2 and 187.5 on two other boxes with the round's first profiled library, `profiles/r6_t1_bench.json`) = %.0f %% of the 16 B/sample roofline, %s with every payload delivered to the host, ragged traffic %s, 30 dB AWGN %s, the multi-GPU code path on one GPU %s (%.2f x); M = 48 %s; 8 channels %s (%s-%s) / %s (pushes of 100 / 400 frames) / **%s with the K = 7 code** (driver r5: 47.7); **64 channels M = 256 QAM16 Golay behind the resampler %s** (driver r5: 88.7); **512 channels behind the oversampled front end %s** (new); 256 channels full duplex %s each way.  CPU oracle %.4f (one thread) / %.3f (16) Gsample/s on the same slab, frames and symbols equal (5.1e-6) |" % (
    g(d["value"]), d["value"] * 1e6 * 16 / 8e12 * 100, g(d["value_with_harvest"]), g(d["value_aperiodic"]), g(d["value_awgn30"]), g(pipe["value"]), pipe["value"] / d["value"], g(m48["value"]),
    g(c1["value"]), g(c1["value_min"]), g(c1["value_max"]), g(c1l["value"]), g(c1v["value"]), g(c2["value"]), g(pf["value"]), g(dx["value"]), d["cpu_baseline"]["value"] / 1000, d["cpu_baseline"]["all_cores"]["value"] / 1000))
s = re.sub(r"58\.9 -> \d+\.\d Gsample/s\*\*", "58.9 -> %s Gsample/s**" % g(pf["value"]), s)
open(p, "w").write(s)

# ---------------- BASELINE.md section 3
p = os.path.join(ROOT, "BASELINE.md"); s = open(p).read()
s = re.sub(r"profiles/r6_t\d_(bench\.json|kernel_stats\.csv|kernel_stats_serial\.csv|traffic\.json|bench_pipeline\.json)", lambda m: "profiles/r6_%s_%s" % (tag, m.group(1)), s)
s = re.sub(r"`r6_t\d_pmc\.csv`", "`r6_%s_pmc.csv`" % tag, s)
s = re.sub(r"runs at \d+\.\d = \d\.\d\d x the direct path", "runs at %s = %.2f x the direct path" % (g(pipe["value"]), pipe["value"] / d["value"]), s)
s = re.sub(r"(\| C2 8-ch M=64 QPSK, pushes of 100 frames per channel \(13\.5 M samples\) \| 1 \| )[^|]*\| [^|]*\|", lambda m: m.group(1) + "%s; driver r5: 81 700 | %.1f %% of 16 B/sample |" % (rng(c1), c1["frac_of_roofline"] * 100), s)
s = re.sub(r"(\| C2 in pushes of 400 frames per channel \(54 M samples\) \| 1 \| )[^|]*\| [^|]*\|", lambda m: m.group(1) + "%s | %.1f %% |" % (rng(c1l), c1l["frac_of_roofline"] * 100), s)
s = re.sub(r"(\| C2 with the K=7 r=1/2 code \| 1 \| )[^|]*\| [^|]*\|", lambda m: m.group(1) + "**%s**; driver r5: 47 700 | %.1f %% |" % (rng(c1v), c1v["frac_of_roofline"] * 100), s)
s = re.sub(r"(\| C2' 512-ch at the reference applications' default numerology M=48 cp=6 \| 1 \| )[^|]*\| [^|]*\|", lambda m: m.group(1) + "%s | %.1f %% |" % (rng(m48), m48["frac_of_roofline"] * 100), s)
s = re.sub(r"(\| \*\*C3\*\* 64-ch M=256 QAM16 Golay\(24,12\) \+ msresamp\(0\.5\) \| 1 \| )[^|]*\| [^|]*\|", lambda m: m.group(1) + "**%s** resampler-input samples (%s - %s); driver r5: 88 677 | %.1f %% of 20 B/sample |" % (sp(c2["value"]), sp(c2["value_min"]), sp(c2["value_max"]), c2["frac_of_roofline"] * 100), s)
s = re.sub(r"(\| C4 512-ch on one GPU \| 1 \| )[^|]*\| [^|]*\| [^|]*\| [^|]*\| [^|]*\|", lambda m: m.group(1) + "%s (%s - %s); delivered to the host %s; ragged traffic %s; 30 dB AWGN %s; without `GPU_MAX_HW_QUEUES=8` %s | %.1f %% of 16 B/sample; %.0f G vector instructions/s = %.0f %% of the issue peak | channelizer %.2f TB/s algorithmic = %.1f %% (%.3f ms alone), 2.60 GB moved per launch = 1.04 x | %.2f | %.1f (16) |" % (
    sp(d["value"]), sp(d["value_min"]), sp(d["value_max"]), sp(d["value_with_harvest"]), sp(d["value_aperiodic"]), sp(d["value_awgn30"]), sp(d["value_default_hw_queues"]),
    d["value"] * 1e6 * 16 / 8e12 * 100, vi["rate_G_per_s"], vi["frac"] * 100, r["achieved"] / 1000, r["frac"] * 100, r["ms_per_launch"], d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"]), s)
s = re.sub(r"(\| \*\*C4' 512-ch behind the channelizer `north_star` names\*\*[^|]*\| 1 \| )[^|]*\| [^|]*\| [^|]*\|", lambda m: m.group(1) + "**%s** (%s - %s); the same chain stage by stage (`front_end = 2`, the round-5 form): 58 921 | %.1f %% of 16 B/sample | %.2f TB/s = **%.1f %%** on 12 B/sample (%.3f ms alone by HIP events in the line; rocprofv3 0.763 = 40.9 %%), 2.67 GB moved = 1.07 x (`profiles/r6_pfb2_traffic.json`) |" % (
    sp(pf["value"]), sp(pf["value_min"]), sp(pf["value_max"]), pf["frac_of_roofline"] * 100, pf["roofline"]["achieved"] / 1000, pf["roofline"]["frac"] * 100, pf["roofline"]["kernel_ms_alone"]), s)
s = re.sub(r"(\| C5 256-ch full duplex on one GPU \| 1 \| )[^|]*\| [^|]*\|", lambda m: m.group(1) + "%s each way as a leg of the driver's line; `bench_duplex.py` in a process of its own: %s (`profiles/r6_%s_duplex.json`) | %.1f %% of 28 B/sample |" % (sp(dx["value"]), sp(dup["value"]), tag, dx["frac_of_roofline"] * 100), s)
open(p, "w").write(s)
print("tables filled from profiles/r6_%s_bench.json: value %s" % (tag, g(d["value"])))
