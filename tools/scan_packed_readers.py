"""Scan hipcc gfx950 assembly for packed-f32 VALU instructions (v_pk_*_f32) that read a register an MFMA wrote fewer
than `limit` wait states earlier (s_nop N counts N + 1, every other instruction 1).  Vector registers AND accumulation
registers are tracked (v_accvgpr_read carries the write time of its source into its destination, so a packed reader
behind such a copy is seen too).  Distances are fall-through distances; at a label the scan keeps what it knows (a
taken back edge can only be LONGER than the fall-through path it skips, or shorter by the loop body it jumps over --
so at every branch target the pending MFMA writes are also aged by zero, i.e. kept: a lower bound on the distance
along the straight path, and the loop-carried case is covered by scanning the body twice).  See rime_common.h
(RIME_MFMA_SETTLE): the failing build of round 2 had 46 such readers, the closest at 13 wait states; builds that pass
have none within 30.  `limit` 24 = the 16 wait states RIME_MFMA_SETTLE guarantees + the >= 8 the compiler's own hazard
recogniser places between an 8-pass MFMA and a VALU read of its result -- an empirical margin, not an ISA number.
--no-packed=SUBSTRING (round 5): kernels whose mangled name contains SUBSTRING must hold NO v_pk_{add,mul,fma}_f32 at all -- the
conjugate-pair kernels, whose blocks share a CU: packed f32 arithmetic of one block beside the MFMA stream of another gave
wrong results there (csrc/fringe_mfma.hip, keep_scalar).
usage: python tools/scan_packed_readers.py file.s [limit] [--fail] [--no-packed=SUBSTRING]     (--fail: exit 1 on a hit)"""
import re, sys
argv = [a for a in sys.argv[1:] if not a.startswith('--')]
path, limit = argv[0], int(argv[1]) if len(argv) > 1 else 30
fail = '--fail' in sys.argv
no_packed = [a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--no-packed=')]
rng = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')


def regs(tok):
    out = set()
    for m in rng.finditer(tok):
        if m.group(1):
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def scan_function(lines):
    """lines: instruction strings of one function.  Walk them twice so that a reader at the top of a loop body sees the
    MFMA at its bottom (time keeps running across the second pass)."""
    t, last, hits = 0, {}, []
    for rnd in range(2):
        for s in lines:
            op = s.split()[0]
            if op == 's_nop':
                t += int(s.split()[1]) + 1
                continue
            t += 1
            args = s[len(op):].split(',')
            if op.startswith('v_mfma') or op.startswith('v_smfmac'):
                for r in regs(args[0]):
                    last[r] = t
                continue
            if op.startswith('v_pk_') and op.endswith('_f32'):
                for r in set().union(*[regs(a) for a in args[1:]]):
                    if r in last and t - last[r] < limit:
                        hits.append((t - last[r], s))
                for r in regs(args[0]):
                    last.pop(r, None)
            elif op.startswith('v_accvgpr_read') or op.startswith('v_accvgpr_mov'):
                src = regs(args[1]) if len(args) > 1 else set()
                w = max((last[r] for r in src if r in last), default=None)
                for r in regs(args[0]):
                    if w is None:
                        last.pop(r, None)
                    else:
                        last[r] = w           # the copy is as fresh as the MFMA result it carries
            elif args and op.startswith(('v_', 'ds_read', 'ds_load', 'global_load', 'buffer_load', 'scratch_load', 'flat_load')):
                for r in regs(args[0]):
                    last.pop(r, None)            # overwritten by a non-MFMA instruction
        if rnd == 0 and not any(l.startswith(('s_cbranch', 's_branch')) for l in lines):
            break                                # no loop: one pass is the whole story
    return sorted(set(hits))


funcs, cur = {}, None
for line in open(path):
    s = line.split(';')[0].split('//')[0].strip()
    if not s:
        continue
    head = s.split()[0]
    if head.endswith(':'):
        if head.startswith('_Z') or not head.startswith(('.', 'BB')):
            cur = funcs.setdefault(head[:-1], [])
        continue
    if s.startswith('.') or cur is None:
        continue
    cur.append(s)
nfunc = nmfma = 0
bad = {}
for f, lines in funcs.items():
    if not any(l.startswith('v_mfma') for l in lines):
        continue
    nfunc += 1
    nmfma += sum(l.startswith('v_mfma') for l in lines)
    hs = scan_function(lines)
    if hs:
        bad[f] = hs
for f, hs in bad.items():
    print(str(f)[:90], ':', len(hs), 'packed readers within', limit, 'wait states; closest', hs[0][0], '|', hs[0][1])
if not bad:
    print('no packed-f32 reader of an MFMA result within', limit, 'wait states (%d kernels with %d MFMAs scanned, '
          'vector and accumulation registers)' % (nfunc, nmfma))
packed = {}
for f, lines in funcs.items():
    if any(sub in f for sub in no_packed):
        n = sum(1 for l in lines if re.match(r'v_pk_(add|mul|fma)_f32\b', l))
        if n:
            packed[f] = n
for f, n in packed.items():
    print(str(f)[:90], ':', n, 'packed f32 instructions in a kernel that must have none')
if no_packed and not packed:
    print('no packed f32 instruction in the %d kernels named *%s*' % (sum(1 for f in funcs if any(sub in f for sub in no_packed)), '*, *'.join(no_packed)))
sys.exit(1 if ((bad or packed) and fail) else 0)
