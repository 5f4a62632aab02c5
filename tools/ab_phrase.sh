#!/bin/bash
# Runs ON THE GPU BOX: kernel-time sum of the phrase level's forward + backward (tools/probe_phrase.py, HIP path) per setting of
# the environment given as arguments ("COATTN_SPLIT=3" ...; "-" = defaults): the probe's wall time is host-paced.
export VQA_PRECISION=fast   # developer tools time the tolerance mode train.Trainer runs (modules default to exact)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for setting in "$@"; do
  rm -rf gpurun_out/abp; mkdir -p gpurun_out/abp
  ( [ "$setting" != "-" ] && export $setting; VQA_PHRASE_ONLY=hip rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp -- python3 tools/probe_phrase.py > gpurun_out/abp.log 2>&1 )
  python3 - "$setting" <<'PY'
import csv, glob, sys, re
f = sorted(glob.glob("gpurun_out/abp/**/*kernel_stats.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "at::" not in r["Name"] and "rocclr" not in r["Name"] and "Cijk" not in r["Name"] and "igemm" not in r["Name"].lower() and "miopen" not in r["Name"].lower()]
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e3
parts = " ".join("%s %.0f" % (re.sub(r"void |\(anonymous namespace\)::|[<(].*", "", r["Name"]).replace("_kernel", ""), float(r["TotalDurationNs"]) / 1e3) for r in rows[:8])
print(sys.argv[1], "own kernels, whole probe: %.0f us |" % tot, parts)
PY
done
