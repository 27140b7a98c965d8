"""Memory-instruction skeleton and register ceiling of one kernel in a tools/isa_dump.sh listing (development aid).
usage: python tools/isa_kernel.py /tmp/clipmi_isa/topk.s rescore_pairs_kernel [max-lines]"""
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 80
s = open(src).read()
for m0 in re.finditer(r"\n[0-9a-f]{16} <([^>]*" + re.escape(pat) + r"[^>]*)>:", s):
    name = m0.group(1)
    start = m0.end()
    m = re.search(r"\n[0-9a-f]{16} <", s[start:])
    body = s[start:start + (m.start() if m else len(s) - start)].splitlines()
    seq = []
    for ln in body:
        t = ln.split()
        if not t:
            continue
        op = t[0]
        if op.startswith(("global_load", "global_store", "buffer_", "ds_", "s_cbranch", "s_barrier")):
            seq.append(op)
        elif op.startswith("s_waitcnt"):
            seq.append(" ".join(t[:3]).split("//")[0].strip())
    out, prev, cnt = [], None, 0
    for x in seq:
        if x == prev:
            cnt += 1
        else:
            if prev:
                out.append(f"{prev} x{cnt}")
            prev, cnt = x, 1
    if prev:
        out.append(f"{prev} x{cnt}")
    text = " ".join(body)
    vmax = max([int(x) for x in re.findall(r"\bv(\d+)\b", text)] + [int(x) for x in re.findall(r"\bv\[\d+:(\d+)\]", text)] + [0])
    amax = max([int(x) for x in re.findall(r"\ba(\d+)\b", text)] + [int(x) for x in re.findall(r"\ba\[\d+:(\d+)\]", text)] + [-1])
    print(f"== {name[:110]}\n   {len(body)} instructions, highest v{vmax}, highest a{amax}, scratch: {'scratch_' in text}")
    print("\n".join("   " + o for o in out[:limit]))
