cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/s5/gputests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/s5/bench20.json 2> gpurun_out/s5/bench20.err
for sb in 32768 131072 262144; do python bench_duplex.py --sub-blocks $sb > gpurun_out/s5/duplex_$sb.json 2> gpurun_out/s5/duplex_$sb.err; done
tail -3 gpurun_out/s5/gputests.txt; cat gpurun_out/s5/bench20.json | cut -c1-400; cat gpurun_out/s5/duplex_*.json | cut -c1-260
