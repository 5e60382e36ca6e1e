import os, sys, json, subprocess
# the fused solve + flux kernel without its stores / without its link arithmetic.  Needs:  make -C pythtb_amd/csrc diag
DIAG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pythtb_amd", "libtbk_diag.so")
for ab in ("0", "1", "3", "4"):
    env = dict(os.environ, TBK_ABLATE_GRID=ab, TBK_LIBRARY=DIAG, TBK_PROF_PERIOD="1")
    out = subprocess.run([sys.executable, "bench.py", "--steps", "300", "--warmup", "30", "--preheat-s", "0.3", "--no-cpu-baseline", "--no-check", "--headline-only"],
                         env=env, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print("ablate", ab, {k: round(v["avg_bracket_ms"] * 1e3, 1) for k, v in j["kernels"].items()}, flush=True)
    except Exception as e:
        print("ablate", ab, "failed", out.stdout[-300:], out.stderr[-600:])
