#!/bin/bash
# round 4: segment-parallel acquisition (default library) against the cadence build of commit 0dabc49, same box, same call:
# the whole bench line (headline, harvest, ragged, the configs block) and the rank-of-8 emulation
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4g; mkdir -p $O
for v in ${VARIANTS:-cadence new}; do
  if [ $v = new ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"
  timeout 900 python bench.py --no-cpu --steps 20 --warmup 5 2>$O/bench_$v.err | tail -1 > $O/bench_$v.json
  python - <<PY
import json
d = json.loads(open("$O/bench_$v.json").read())
print("  value", d["value"], d.get("value_min"), d.get("value_max"), "harvest", d.get("value_with_harvest"), "aper", d.get("value_aperiodic"), d.get("value_aperiodic_over_value"))
print("  alone", d["roofline"]["kernels_ms"]); print("  overlapped", d["roofline"]["kernels_ms_overlapped"]); print("  ", d["frames_acquired"], d["verified"]["ok"])
for k, c in (d.get("configs") or {}).items():
    print("  cfg", k, c.get("value"), c.get("frac_of_roofline"), c.get("frames_acquired"), (c.get("verified") or {}).get("ok"), c.get("error"), (c.get("kernels_ms_overlapped") or {}).get("sync_kernel"))
PY
  tail -3 $O/bench_$v.err
  if [ $v = cadence ]; then for e in 0 12; do echo "  rank-of-8 emulation, MCRX_EXTRA_ROUNDS=$e"; MCRX_EXTRA_ROUNDS=$e MG8_WARM=48 python scratch/mg8_stream.py 2 2>&1 | tail -1 | cut -c1-400; done
  else echo "  rank-of-8 emulation"; python scratch/mg8_stream.py 2 2>&1 | tail -1 | cut -c1-400; fi
done
