#!/usr/bin/env python3
"""Summarise the [layer] lines bench.py --layer-table writes to stderr: best plan per layer, class totals."""
import re, sys
tot = 0; cls = {}
for l in open(sys.argv[1]):
    if not l.startswith('[layer]'): continue
    name = l.split()[1]
    ns = re.search(r'us: (.*?) \| split (.*?)  best=(\(.*?\))\s+([\d.]+) TF', l)
    v = dict((k, float(x)) for k, x in re.findall(r'(v\d+)=\s*([\d.]+)', ns.group(1)))
    sp = re.findall(r'(v\d+m\d+g\d+o\d+)=\s*([\d.]+)', ns.group(2))
    bestv = min(v, key=v.get)
    b = min([v[bestv]] + [float(x) for _, x in sp])
    tot += b
    kind = 'stem' if name == 'stem' else name[4:]
    key = name[:2] + ' ' + kind if name != 'stem' else 'stem'
    cls.setdefault(key, []).append((b, float(ns.group(4))))
    if len(sys.argv) > 2:
        print(f"{name:8s} nosplit {bestv}={v[bestv]:6.1f}  split {sp[0] if sp else ''}  best={ns.group(3)} {ns.group(4)} TF/s")
for k in sorted(cls):
    xs = cls[k]
    print(f"{k:8s} n={len(xs):2d} sum={sum(b for b,_ in xs):7.1f} us  avg {sum(t for _,t in xs)/len(xs):5.1f} TF/s")
print("sum best us", round(tot, 1))
