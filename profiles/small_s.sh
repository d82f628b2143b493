#!/bin/bash
# launch-bound regime: minibatches/s at small stream counts and on the arxiv-like graph (run via gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
X="--no-cpu-baseline --no-compat --e2e-steps 0 --no-kernel-timing"
show='import json,sys; L=[l for l in sys.stdin if l.startswith(chr(123))]; d=json.loads(L[-1]); print(sys.argv[1], "%.1f k minibatches/s, %.2f G edges/s, %.3f ms/round" % (d["iters_per_sec"]/1e3, d["value"]/1e9, d["ms_per_step"]))'
for S in 1 4 16 64; do
  python3 bench.py --streams $S --steps 200 --warmup 20 $X 2>/dev/null | python3 -c "$show" "products-like S=$S"
done
python3 bench.py --nodes 169343 --mean-deg 6.9 --fanout 10,10 --streams 128 --steps 200 --warmup 20 $X 2>/dev/null | python3 -c "$show" "arxiv-like 10/10 S=128"
python3 bench.py --nodes 169343 --mean-deg 6.9 --fanout 10,10 --streams 16 --steps 200 --warmup 20 $X 2>/dev/null | python3 -c "$show" "arxiv-like 10/10 S=16"
