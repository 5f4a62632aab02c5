"""Developer probe: forward-kernel time (bench.roofline_leg) for a few environment settings, one process each."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for env in sys.argv[1:] or [""]:
    e = dict(os.environ)
    for kv in env.split(","):
        if kv:
            k, v = kv.split("=")
            e[k] = v
    out = subprocess.run([sys.executable, "-c", "import json,torch,bench; d=torch.device('cuda',0); "
                          "print(json.dumps([bench.roofline_leg(d)['avg_launch_us'], bench.roofline_leg(d, N=49)['avg_launch_us']]))"],
                         cwd=ROOT, env=e, capture_output=True, text=True)
    print(env or "(default)", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
