#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): regenerates everything profiles/ is built from into
# gpurun_out/refresh/.  tools/collect_profiles.py then copies the summaries into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python3 bench.py --no-cpu-baseline > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/roofline -- python3 bench.py --only roofline > $O/roofline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hot -- python3 bench.py --only hot > $O/hot.log 2>&1
# the headline kernel alone (cfg-2 shape, location-major), timed exactly as bench.py's roofline leg times it
ITERS=100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/headline -- python3 tools/probe_fwd_one.py > $O/headline.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/step.log 2>&1
rm -f $O/*/*/*kernel_trace.csv $O/hot/*/*_trace.csv
# counter passes of the headline kernel: the cfg-2 shape, location-major features, nothing else in the process
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 tools/probe_fwd_one.py > $O/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- python3 tools/probe_fwd_one.py > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU --output-format csv -d $O/pmc_hot -- python3 bench.py --only hot > $O/pmc_hot.log 2>&1
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE 160 196 26 512 3 lm > $O/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
tail -c 600 $O/bench.json
