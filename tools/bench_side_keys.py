#!/usr/bin/env python3
"""Print the latency-bound extra keys of a bench line (development aid): python tools/bench_side_keys.py FILE"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f, p, t = d.get("forward", {}), d.get("forward_fpn", {}), d.get("detect", {})
print("step %.0f ms | C4 f32 group4 %s, bf16 group4 %s | FPN f32 alone %s | detect alone %s" % (
    d["ms_per_step"], f.get("ms_per_image_f32_group4"), f.get("ms_per_image_bf16_group4"), p.get("ms_per_image_f32"), t.get("ms_per_image")))
