#!/bin/bash
# kernel trace of the small-S regime (run via gpurun): where a round's 0.37 ms go when S <= 16
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/small_s_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in 1 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$S -- python3 $R/bench.py --streams $S --steps 100 --warmup 10 --no-cpu-baseline --no-compat --no-kernel-timing --e2e-steps 0 > $OUT/s$S.log 2>&1
  python3 $R/profiles/timeline.py $OUT/s$S > $OUT/timeline_s$S.txt
done
cat $OUT/timeline_s1.txt $OUT/timeline_s16.txt
