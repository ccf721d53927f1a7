"""Scan hipcc assembly for packed-f32 VALU instructions (v_pk_*_f32) that read a register an MFMA wrote fewer than
`limit` wait states earlier in straight-line code (s_nop N counts N + 1, every other instruction 1; fall-through
distance, a lower bound across branches).  See rime_common.h (RIME_MFMA_SETTLE).
usage: python tools/scan_packed_readers.py file.s [limit]"""
import re, sys
path, limit = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = re.compile(r'v\[(\d+):(\d+)\]|\bv(\d+)\b')


def regs(tok):
    out = set()
    for m in rng.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


func, t, last = None, 0, {}
hits = {}
for line in open(path):
    s = line.strip()
    if s.startswith('_Z') and ':' in s.split()[0]:
        func, t, last = s.split(':')[0], 0, {}
        continue
    if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
        continue
    op = s.split()[0]
    if op == 's_nop':
        t += int(s.split()[1]) + 1
        continue
    t += 1
    args = s[len(op):].split(',')
    if op.startswith('v_mfma'):
        for r in regs(args[0]):
            last[r] = t
        continue
    if op.startswith('v_pk_') and op.endswith('_f32'):
        for r in set().union(*[regs(a) for a in args[1:]]):
            if r in last and t - last[r] < limit:
                hits.setdefault(func, []).append((t - last[r], s))
    elif args and op.startswith('v_'):
        for r in regs(args[0]):
            last.pop(r, None)            # overwritten by a non-MFMA instruction
for f, hs in hits.items():
    hs.sort()
    print(str(f)[:90], ':', len(hs), 'packed readers within', limit, 'wait states; closest', hs[0][0], '|', hs[0][1])
if not hits:
    print('no packed-f32 reader of an MFMA result within', limit, 'wait states')
