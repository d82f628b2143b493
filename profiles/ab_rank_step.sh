cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in 1024 128; do
echo -n "rank path (RCCL world 1), batch $b, native: "; python3 profiles/e2e_only.py --rank-path --batch $b --steps 256 2>/dev/null | tail -1 | cut -c60-125
echo -n "rank path (RCCL world 1), batch $b, python: "; CSLICER_PY_STEP=1 python3 profiles/e2e_only.py --rank-path --batch $b --steps 256 2>/dev/null | tail -1 | cut -c60-125
done; done
