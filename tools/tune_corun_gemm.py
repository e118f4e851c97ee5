#!/usr/bin/env python3
"""The GEMM launch plans of the BERT-base encoder chosen under self-co-run (round 6; see tools/tune_corun.py): the default schedule of tools/bench_bert.py runs four
replicas side by side.  Each of the four projection shapes runs on `--lanes` streams at once (captured graphs of REPS launches, activation-like random operands, bias and
-- for the first feed-forward product -- the fused Gelu), and the figure is the time per launch over all streams.
    python tools/tune_corun_gemm.py [--lanes 4] [--out profiles/plans/experiments/bert_corun4.json]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # every stream a hardware queue of its own, as in bench.py (must be set before the HIP runtime starts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=4)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--plan", default=os.path.join(ROOT, "profiles", "plans", "bert_base_b32_s128.json"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "plans", "experiments", "bert_corun4.json"))
    args = ap.parse_args()
    from rten_amd import lib as L
    from rten_amd.tensor import DeviceTensor
    incumbent = json.load(open(args.plan))
    bkey = next(iter(incumbent))
    table = incumbent[bkey]
    ctxs = [L.Context(0) for _ in range(args.lanes)]
    rng = np.random.default_rng(3)
    M = 4096
    shapes = {"ctx.qkv": (768, 2304, L.ACT_NONE), "o.matmul": (768, 768, L.ACT_NONE), "ffn1.matmul": (768, 3072, L.ACT_GELU), "ffn2.matmul": (3072, 768, L.ACT_NONE)}
    out_table = dict(table)
    tot_inc = tot_new = 0.0
    for suffix, (K, N, act) in shapes.items():
        names = [n for n in table if n.endswith(suffix)]
        inc = list(table[names[0]])
        ops = []
        for c in ctxs:
            a = DeviceTensor.from_numpy(c, rng.standard_normal((M, K), dtype=np.float32))
            w = DeviceTensor.from_numpy(c, (rng.standard_normal((K, N), dtype=np.float32) * 0.03).astype(np.float32))
            b = DeviceTensor.from_numpy(c, rng.standard_normal(N, dtype=np.float32))
            o = DeviceTensor(c, (M, N), np.float32)
            ops.append((c, a, w, b, o))
        d = L.gemm_desc(M, N, K, K, 1, N, 1, N, bias_kind=L.BIAS_PER_COL, act=act)

        def launch(c, a, w, b, o, plan):
            v, mode, groups, order = plan
            c.set_gemm_variant(v); c.call("rten_hip_set_gemm_split", mode, groups); c.call("rten_hip_set_gemm_order", order)
            c.call("rten_hip_gemm_f32", C.byref(d), a.vp, w.vp, b.vp, o.vp)
            c.set_gemm_variant(-1); c.call("rten_hip_set_gemm_split", 3, 1); c.call("rten_hip_set_gemm_order", 0)

        def measure(plan):
            graphs = []
            try:
                for op in ops:
                    launch(*op, plan)
                for c in ctxs:
                    c.sync()
                for op in ops:
                    op[0].graph_begin()
                    for _ in range(args.reps):
                        launch(*op, plan)
                    graphs.append((op[0], op[0].graph_end()))
                best = 1e30
                for _ in range(3):
                    t0 = time.perf_counter()
                    for c, g in graphs:
                        c.graph_launch(g)
                    for c in ctxs:
                        c.sync()
                    best = min(best, (time.perf_counter() - t0) / (args.reps * len(ops)) * 1e6)
                return best
            finally:
                for c, g in graphs:
                    c.graph_destroy(g)

        nblk = (K + 255) // 256
        cands = [[v, 3, 1, o] for v in (0, 1, 2, 3, 12, 13, 14, 15, 16, 17, 18, 19, 27) for o in (0, 1)]
        cands += [[v, 0, 1, o] for v in (0, 1, 2, 3, 27) for o in (0, 1)]
        cands += [[v, 5, r, o] for v in (0, 1, 2, 3) for r in (1, 2, 3) for o in (0, 1)]
        for g in sorted({2, 3, 4, 6, nblk} & set(range(2, nblk + 1))):
            cands += [[v, m, g, o] for v in (0, 1, 2, 3, 27) for (m, o) in ((1, 0), (2, 0), (2, 3))]
        t_inc = min(measure(inc), measure(inc))
        rows = []
        for cand in cands:
            if cand == inc:
                continue
            try:
                rows.append((measure(cand), cand))
            except L.HipError:
                continue
        rows.sort()
        best_t, best_c = t_inc, inc
        for t, cand in rows[:3]:
            t2 = min(t, measure(cand))
            if t2 < best_t * 0.99:
                best_t, best_c = t2, cand
        fl = 2.0 * M * K * N
        print(f"{suffix:12s} {M}x{K}x{N}: incumbent {inc} {t_inc:7.1f} us ({fl / t_inc / 1e6:5.1f} TF/s) -> {best_c} {best_t:7.1f} us ({fl / best_t / 1e6:5.1f}) | " +
              " ".join(f"{c}={t:.1f}" for t, c in rows[:8]), flush=True)
        for n in names:
            out_table[n] = best_c
        tot_inc += t_inc * len(names); tot_new += best_t * len(names)
    print(f"# sum over the encoder's products: incumbent {tot_inc:.0f} us, chosen {tot_new:.0f} us")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({bkey: out_table}, open(args.out, "w"))


if __name__ == "__main__":
    main()
