# the acquisition rounds' scout: budgeted build against the unbudgeted one (MCRX_LEAN_BUILD), 8-channel stream and the headline
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/lean; mkdir -p $O
for lb in 0 1; do
  MCRX_LEAN_BUILD=$lb python scratch/configs_r2.py C2 C2_conv_v27 C3 2>/dev/null | tail -1 > $O/cfg_$lb.json
  MCRX_LEAN_BUILD=$lb python bench.py --no-cpu --no-harvest --no-aperiodic --steps 30 --warmup 8 --serial-steps 2 2>/dev/null | tail -1 > $O/bench_$lb.json
  python - <<PY
import json
d=json.loads(open("$O/cfg_$lb.json").read())
for k,v in d.items(): print("lean_build $lb", k, v["Msamples_per_s"], v["ms_per_step"], v["kernels_ms_overlapped"], v["verified"])
b=json.loads(open("$O/bench_$lb.json").read())
print("lean_build $lb bench", b["value"], b["verified"]["ok"], b["roofline"]["kernels_ms_overlapped"])
PY
done
