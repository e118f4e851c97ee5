#!/usr/bin/env python3
"""sdpa_fused16_kernel at BERT-base's shape under a start delay per co-resident workgroup class (RTEN_SDPA_STAGGER = 10 ns ticks; tuning only):
python tools/probe_sdpa16.py [ticks ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import probe_sdpa  # noqa: E402

import subprocess
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    r = probe_sdpa.run(int(sys.argv[2]))
    print(f"mask {r['mask']:6.2f} us   no mask {r['no mask']:6.2f} us")
else:
    for ticks in [int(v) for v in sys.argv[1:]] or [0, 100, 200, 300, 400]:
        for dbg in (0,):
            env = dict(os.environ, RTEN_SDPA_STAGGER=str(ticks))
            out = subprocess.run([sys.executable, __file__, "--one", str(dbg)], env=env, capture_output=True, text=True).stdout.strip()
            print(f"stagger {ticks * 10:5d} ns per class: {out}", flush=True)
    env = dict(os.environ)
    out = subprocess.run([sys.executable, __file__, "--one", str(0x200000)], env=env, capture_output=True, text=True).stdout.strip()
    print(f"32-query form:               {out}", flush=True)
