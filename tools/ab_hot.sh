#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of the co-attention forward + backward at N = 196 and 49, for each setting of
# the environment given as arguments ("COATTN_SPLIT=3" "COATTN_SPLIT_FWD=2" ...; "-" = defaults).
export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab_hot
mkdir -p $O
for setting in "$@"; do
  tag=$(echo "$setting" | tr ' =' '__')
  for N in 196 49; do
    find gpurun_out/ab_hot -name "*_stats.csv" -delete 2>/dev/null       # (a directory is reused from run to run: kstats.py takes the first file)
    ( [ "$setting" != "-" ] && export $setting; rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_$N -- python3 tools/probe_hot.py $N ${LAYOUT:-lm} 200 > $O/${tag}_$N.log 2>&1 )
    rm -f $O/${tag}_$N/*/*kernel_trace.csv
    echo "== $setting N=$N: $(tail -1 $O/${tag}_$N.log)"
    python3 tools/kstats.py $O/${tag}_$N 14
  done
done
