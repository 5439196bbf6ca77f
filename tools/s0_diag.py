"""Developer diagnostic: stage0 megakernel time with phases switched off (results are wrong then)."""
import os, sys, subprocess, json
batch = sys.argv[1] if len(sys.argv) > 1 else "1024"
for diag in (0, 64, 51, 115):
    env = dict(os.environ, BTSBOT_AMD_S0_DIAG=str(diag))
    r = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--batch", batch],
                         env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout)
        print(f"B={batch} diag={diag}: stage0 {d['kernels']['stage0_kernel']['avg_launch_us']} us  stage1 {d['kernels']['stage1_kernel']['avg_launch_us']} us  step {d['ms_per_step']} ms")
    except Exception:
        print(f"diag={diag}: failed", r.stderr[-200:])
