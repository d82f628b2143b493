#!/bin/bash
# bench.py's three launch forms on ONE GPU (run via gpurun): plain, two ranks rehearsed over gloo (the driver
# uses nccl = RCCL on several GPUs), and the e2e watchdog path forced by a tiny timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
show='import json,sys; L=[l for l in sys.stdin if l.startswith(chr(123))]; d=json.loads(L[-1]); print(len(L), "json line(s): n_gpus", d["n_gpus"], "G edges/s %.2f" % (d["value"]/1e9), "e2e", d.get("e2e", {}).get("iters_per_sec", d.get("e2e")))'
python3 bench.py --steps 10 --no-cpu-baseline --no-compat 2>/dev/null | python3 -c "$show"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --streams 32 2>/dev/null | python3 -c "$show"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --streams 32 --e2e-timeout 0.5 2>/dev/null | python3 -c "$show"
echo "exit code of the last launch: $?"
