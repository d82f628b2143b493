X="--no-cpu-baseline --no-compat --e2e-steps 0 --no-kernel-timing"
show='import json,sys; L=[l for l in sys.stdin if l.startswith(chr(123))]; d=json.loads(L[-1]); print(sys.argv[1], "%.1f k minibatches/s, %.3f ms/round" % (d["iters_per_sec"]/1e3, d["ms_per_step"]))'
for st in 60 800; do
for S in 1 16; do
  python3 bench.py --streams $S --steps $st --warmup 10 $X 2>/dev/null | python3 -c "$show" "products-like S=$S steps=$st"
done
done
