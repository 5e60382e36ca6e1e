#!/usr/bin/env python3
"""config E solve_on_grid at a given mesh size with 1, 2, 3 chunks in flight (TBK_TW16_STREAMS) and against the QL-replay
form (TBK_TW16=0).  python profiles/tw16_streams_probe.py [side]"""
import contextlib, ctypes as C, io, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pythtb_amd as tb
from pythtb_amd import _lib
import helpers as hp
side = int(sys.argv[1]) if len(sys.argv) > 1 else 257
with contextlib.redirect_stdout(io.StringIO()):
    m = hp.cubic16(tb.tb_model)
w = tb.wf_array(m, [side] * 3)
out = {"side": side}
for tag, env in (("streams3", {}), ("streams1", {"TBK_TW16_STREAMS": "1"}), ("streams2", {"TBK_TW16_STREAMS": "2"}),
                 ("ws16g", {"TBK_QLW_WS_MB": "16384"}), ("ws2g", {"TBK_QLW_WS_MB": "2048"}), ("replay", {"TBK_TW16": "0"})):
    with contextlib.ExitStack() as st:
        for k, v in env.items():
            st.enter_context(_lib.knob(k, v))
        w.solve_on_grid([0.0, 0.0, 0.0])
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); g = w.solve_on_grid([0.0, 0.0, 0.0]); ts.append(time.perf_counter() - t0)
        out[tag] = {"ms": 1e3 * min(ts), "kpts_per_s": (side - 1) ** 3 / min(ts), "gap78": float(g[7])}
print(json.dumps(out))
