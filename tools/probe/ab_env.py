"""Same-box A/B of a bench.py record under several environments, alternating:  python tools/probe/ab_env.py ROUNDS "bench args" "ENV1" "ENV2" ...
(ENV = space-separated K=V pairs, '-' for none); prints ms_per_step of every run."""
import json, os, subprocess, sys
rounds, args, envs = int(sys.argv[1]), sys.argv[2].split(), sys.argv[3:]
for r in range(rounds):
    for e in envs:
        env = dict(os.environ)
        env.update(kv.split("=", 1) for kv in e.split() if kv != "-")
        out = subprocess.run([sys.executable, "bench.py"] + args, env=env, capture_output=True, text=True).stdout.strip().splitlines()
        try:
            d = json.loads(out[-1])
            print(f"{e:40s} {d['ms_per_step']}", flush=True)
        except Exception as ex:
            print(f"{e:40s} FAILED {ex}", flush=True)
