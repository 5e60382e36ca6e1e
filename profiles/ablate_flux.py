import os, sys, json, subprocess
for ab in ("0", "1", "2"):
    env = dict(os.environ, TBK_ABLATE_FLUX=ab)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "3", "--no-cpu-baseline", "--no-check"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("ablate_flux", ab, {k: round(v["avg_ms"]*1e3,1) for k, v in j["kernels"].items()})
    except Exception as e:
        print("ablate_flux", ab, "failed", out.stdout[-300:], out.stderr[-600:])
