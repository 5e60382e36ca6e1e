#!/usr/bin/env python3
"""profiles/traffic.json (what bench.py quotes as roofline.traffic) = the per-directory traffic.json files of the current
collections merged, first directory wins per kernel:   python profiles/merge_traffic.py r04b r04c r04bcfg r04f"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) < 2:
    sys.exit("merge_traffic.py: name at least one collection directory (an empty run would clobber traffic.json)")
out = {}
for d in sys.argv[1:]:
    for k, v in json.load(open(os.path.join(HERE, d, "traffic.json"))).items():
        if k.startswith("__amd") or k in out:
            continue
        out[k] = v
if not out:
    sys.exit("merge_traffic.py: no kernels found in %s; traffic.json left as it was" % sys.argv[1:])
out["__sources__"] = sys.argv[1:]
json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1, sort_keys=True)
print(len(out), "kernels")
