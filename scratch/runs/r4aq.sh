#!/bin/bash
# txsym64_kernel (eight symbols side by side, eight points per lane) against the one-point-per-lane kernel (MCTX_TXSYM64=0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4aq; mkdir -p $O
cd $R; timeout 1200 python -m pytest tests -m gpu -x -q -k "tx or refapp or ofdm or duplex or baseline" 2>&1 | tail -3; cd /tmp
for v in 0 1; do
  MCTX_TXSYM64=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O -o v$v -- python $R/bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 2 --warmup 1 --reps 1 --serial-steps 1 > $O/v$v.log 2>&1
  echo "== MCTX_TXSYM64=$v"; grep -E "txsym" $O/v${v}_kernel_stats.csv; tail -1 $O/v$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  verified', d['verified'])"
done
for v in 0 1 0 1; do
  echo "== duplex MCTX_TXSYM64=$v"; MCTX_TXSYM64=$v python $R/bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], d['ms_per_step'], d['verified']['ok'])"
done
