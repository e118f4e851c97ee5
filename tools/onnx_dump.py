#!/usr/bin/env python3
"""Lists the nodes of an ONNX file (no `onnx` package in this image: a minimal protobuf reader of ModelProto / GraphProto / NodeProto / TensorProto).
    python tools/onnx_dump.py model.onnx [--skip-ops MatMul,Add]      # op, name, inputs -> outputs; small int tensors are printed
Tooling only: nothing under rten_amd/ imports this."""
import struct
import sys


def varint(b, i):
    v = s = 0
    while True:
        c = b[i]; i += 1
        v |= (c & 0x7F) << s
        if not c & 0x80:
            return v, i
        s += 7


def fields(b):
    i = 0
    while i < len(b):
        k, i = varint(b, i)
        f, w = k >> 3, k & 7
        if w == 0:
            v, i = varint(b, i)
        elif w == 1:
            v, i = b[i:i + 8], i + 8
        elif w == 2:
            n, i = varint(b, i)
            v, i = b[i:i + n], i + n
        elif w == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f"wire type {w}")
        yield f, w, v


def tensor(b):
    t = {"dims": [], "type": 0, "name": "", "raw": b"", "i64": [], "i32": [], "f32": []}
    for f, w, v in fields(b):
        if f == 1:
            t["dims"] += [v] if w == 0 else [x for x in _packed(v)]
        elif f == 2:
            t["type"] = v
        elif f == 8:
            t["name"] = v.decode()
        elif f == 9:
            t["raw"] = bytes(v)
        elif f == 7:
            t["i64"] += [v] if w == 0 else list(_packed(v))
        elif f == 5:
            t["i32"] += [v] if w == 0 else list(_packed(v))
        elif f == 4:
            t["f32"] += list(struct.unpack(f"<{len(v) // 4}f", v))
    return t


def _packed(v):
    i = 0
    while i < len(v):
        x, i = varint(v, i)
        yield x if x < (1 << 63) else x - (1 << 64)


def tensor_values(t, limit=16):
    n = 1
    for d in t["dims"]:
        n *= d
    if n > limit:
        return None
    if t["type"] == 7:
        return list(struct.unpack(f"<{len(t['raw']) // 8}q", t["raw"])) if t["raw"] else t["i64"]
    if t["type"] == 6:
        return list(struct.unpack(f"<{len(t['raw']) // 4}i", t["raw"])) if t["raw"] else t["i32"]
    if t["type"] == 1:
        return list(struct.unpack(f"<{len(t['raw']) // 4}f", t["raw"])) if t["raw"] else t["f32"]
    if t["type"] in (2, 3, 9):
        return list(t["raw"]) if t["raw"] else t["i32"]
    return None


def attr(b):
    a = {"name": "", "i": None, "f": None, "s": None, "ints": [], "t": None}
    for f, w, v in fields(b):
        if f == 1:
            a["name"] = v.decode()
        elif f == 2:
            a["f"] = struct.unpack("<f", v)[0]
        elif f == 3:
            a["i"] = v if v < (1 << 63) else v - (1 << 64)
        elif f == 4:
            a["s"] = v.decode(errors="replace")
        elif f == 5:
            a["t"] = tensor(v)
        elif f == 8:
            a["ints"] += [v if v < (1 << 63) else v - (1 << 64)] if w == 0 else list(_packed(v))
    return a


def node(b):
    n = {"in": [], "out": [], "name": "", "op": "", "attrs": []}
    for f, w, v in fields(b):
        if f == 1:
            n["in"].append(v.decode())
        elif f == 2:
            n["out"].append(v.decode())
        elif f == 3:
            n["name"] = v.decode()
        elif f == 4:
            n["op"] = v.decode()
        elif f == 5:
            n["attrs"].append(attr(v))
    return n


def load(path):
    data = open(path, "rb").read()
    nodes, inits = [], {}
    for f, w, v in fields(data):
        if f == 7:
            for f2, w2, v2 in fields(v):
                if f2 == 1:
                    nodes.append(node(v2))
                elif f2 == 5:
                    t = tensor(v2)
                    inits[t["name"]] = t
    return nodes, inits


def main():
    nodes, inits = load(sys.argv[1])
    skip = set(sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[2] == "--skip-ops" else set()
    consts = {}
    for n in nodes:
        if n["op"] == "Constant":
            for a in n["attrs"]:
                if a["t"] is not None:
                    consts[n["out"][0]] = a["t"]
    for n in nodes:
        if n["op"] in skip or n["op"] == "Constant":
            continue

        def show(v):
            t = consts.get(v) or inits.get(v)
            if t is None:
                return v
            vals = tensor_values(t)
            return f"{v}={vals}" if vals is not None else f"{v}<{t['dims']}>"
        at = " ".join(f"{a['name']}={a['i'] if a['i'] is not None else a['ints'] or a['f'] or a['s']}" for a in n["attrs"] if a["t"] is None)
        print(f"{n['op']:18s} {', '.join(show(v) for v in n['in'])}  ->  {', '.join(n['out'])}   {at}")


if __name__ == "__main__":
    main()
