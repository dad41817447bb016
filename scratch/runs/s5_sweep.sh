# slab-size sweep of the headline bench on the current build (frames per channel and slab), same box, same clocks
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/sweep; mkdir -p $O
for f in 16 24 32 16; do
  python bench.py --frames $f --no-cpu --no-harvest --no-aperiodic --steps 30 --warmup 8 --serial-steps 2 2>/dev/null | tail -1 > $O/f$f.json
  python - <<PY
import json
d=json.loads(open("$O/f$f.json").read())
print("frames", $f, "value", d["value"], "ms/step", d["ms_per_step"], "ok", d["verified"]["ok"], d["roofline"]["kernels_ms_overlapped"])
PY
done
