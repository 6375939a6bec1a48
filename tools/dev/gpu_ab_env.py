"""A/B of an environment switch the library reads at first use (one fresh process per value): graph-replayed step time of the bench workload.
    python tools/dev/gpu_ab_env.py NAME v0 v1 ...   [HARP_AB_STEPS=60]"""
import os, subprocess, sys
name, vals = sys.argv[1], sys.argv[2:]
here = os.path.dirname(os.path.abspath(__file__))
for rep in range(2):
    for v in vals:
        env = dict(os.environ); env[name] = v
        r = subprocess.run([sys.executable, os.path.join(here, "gpu_step_ms.py")], env=env, capture_output=True, text=True)
        print(f"{name}={v}:", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
