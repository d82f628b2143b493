#!/bin/bash
# Records csl_gemm_f32 plans for the SMALL size buckets (a 128-seed rank / replica step) and for the attention model's
# classes (run on the GPU box via gpurun): every solution of the library is timed per shape class (CSLICER_GEMM_TUNE=all); the union with the
# shipped file goes to gpurun_out/gemm_plans_gfx950.txt (copy to occ-gnn_amd/cslicer/ to ship it).
cd ${GRAFT_REPO_ROOT:-.}
for mode in single rank gat gat128; do
CSLICER_GEMM_TUNE=all MODE=$mode python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os, sys
sys.path.insert(0, "occ-gnn_amd")
mode = os.environ["MODE"]
sys.argv = ["e2e_only.py", "--steps", "64", "--warmup", "32", "--streams", "32"]
if mode in ("single", "rank"):
    sys.argv += ["--batch", "128"] + (["--rank-path"] if mode == "rank" else [])
else:   # the attention model's classes (config 5's shape), batch 1024 and a 128-seed replica
    sys.argv += ["--model", "gat", "--fanout", "10,10,10", "--hidden", "32"] + (["--batch", "128"] if mode == "gat128" else [])
try:
    exec(open("profiles/e2e_only.py").read())
finally:
    from cslicer import aggr
    aggr.gemm_save_plans("gpurun_out/gemm_plans_%s.txt" % os.environ["MODE"])
PY
done
python3 - <<'PY'
import os
keep, order = {}, []
hdr = None
for f in ("occ-gnn_amd/cslicer/gemm_plans_gfx950.txt", "gpurun_out/gemm_plans_single.txt", "gpurun_out/gemm_plans_rank.txt",
          "gpurun_out/gemm_plans_gat.txt", "gpurun_out/gemm_plans_gat128.txt"):
    if not os.path.exists(f):
        continue
    for line in open(f):
        if line.startswith("#"):
            hdr = hdr or line
            continue
        p = line.split()
        if len(p) != 14:
            continue
        k = tuple(p[:13])
        if k not in keep:
            keep[k] = p[13]
            order.append(k)
with open("gpurun_out/gemm_plans_gfx950.txt", "w") as o:
    o.write(hdr)
    for k in order:
        o.write(" ".join(k) + " " + keep[k] + "\n")
print(open("gpurun_out/gemm_plans_gfx950.txt").read())
PY
