"""Developer diagnostic: head kernel time with parts switched off (results are wrong then)."""
import os, sys, subprocess, json
for diag in (0, 1, 2, 4, 8, 16, 31):
    env = dict(os.environ, BTSBOT_AMD_HEAD_DIAG=str(diag), BTSBOT_AMD_S0_DIAG="0")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout)
        print(f"diag={diag}: head {d['kernels']['head_kernel']['avg_launch_us']} us")
    except Exception:
        print(f"diag={diag}: failed", r.stderr[-300:])
