#!/bin/bash
# round 2, first GPU pass: parity tests on the pipelined engine, then the new bench in both receiver modes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_first; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --serial --no-cpu --no-harvest > $O/bench_serial.json 2> $O/bench_serial.err; echo "serial rc=$?"
timeout 300 python bench.py --no-cpu --no-harvest --steps 20 --warmup 5 > $O/bench_short.json 2> $O/bench_short.err; echo "short rc=$?"
tail -c 3000 $O/bench.json; tail -5 $O/bench.err
tail -c 1500 $O/bench_serial.json
tail -c 1500 $O/bench_short.json
