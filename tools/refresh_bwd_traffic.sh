export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; mkdir -p gpurun_out/refresh
for N in 49 196; do
  for c in FETCH_SIZE WRITE_SIZE; do
    find $O/pmcb_${c}_$N -name "*.csv" -delete 2>/dev/null
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmcb_${c}_$N -- python3 tools/probe_hot.py $N lm 30 > $O/pmcb_${c}_$N.log 2>&1
  done
  python3 tools/pmc_traffic_bwd.py $O/pmcb_FETCH_SIZE_$N $O/pmcb_WRITE_SIZE_$N 160 $N 26 512 3 lm | cut -c1-600
done
cp profiles/pmc_traffic_backward.json $O/pmc_traffic_backward.json
