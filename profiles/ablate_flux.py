import os, sys, json, subprocess
# needs the diagnostic build:  make -C pythtb_amd/csrc diag   (the shipped libtbk.so carries no ablation code)
DIAG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pythtb_amd", "libtbk_diag.so")
for ab in ("0", "1", "2"):
    env = dict(os.environ, TBK_ABLATE_FLUX=ab, TBK_LIBRARY=DIAG)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "3", "--no-cpu-baseline", "--no-check"], env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("ablate_flux", ab, {k: round(v["avg_ms"]*1e3,1) for k, v in j["kernels"].items()})
    except Exception as e:
        print("ablate_flux", ab, "failed", out.stdout[-300:], out.stderr[-600:])
