#!/usr/bin/env python3
"""Developer tool: a one-line-per-basic-block trace of instruction classes of one kernel in an assembly file (-S output):
M mfma, BL1/BL4 buffer loads, GL global loads, D LDS, ST stores, W(..) waits, BAR barriers, v<N> runs of other VALU.
usage: tools/isa_trace.py file.s <substring of the mangled kernel name>"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
cands = [i for i, l in enumerate(txt) if l.startswith('_Z') and sys.argv[2] in l.split(':')[0] and ':' in l]
if not cands:
    raise SystemExit("no kernel whose mangled name contains %r (kernels: grep '^_Z.*:' %s)" % (sys.argv[2], sys.argv[1]))
start = cands[0]
out = []
valu = 0
def flush():
    global valu
    if valu: out.append('v%d' % valu); valu = 0
for l in txt[start + 1:]:
    t = l.strip()
    if t.startswith('.Lfunc_end') or t.startswith('.end_amdhsa_kernel'): break
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'): continue
    op = t.split()[0]
    if t.startswith('.LBB'): flush(); out.append('\n' + t.split(':')[0] + ':')
    elif op.startswith('s_cbranch') or op == 's_branch': flush(); out.append(op[2:] + '->' + t.split()[1])
    elif op.startswith('buffer_load'): flush(); out.append('BL4' if 'x4' in op else 'BL1')
    elif op.startswith('global_load'): flush(); out.append('GL')
    elif op.startswith('scratch_'): flush(); out.append('SCR')
    elif op.startswith('v_mfma'): flush(); out.append('M')
    elif op == 's_waitcnt': flush(); out.append('W(' + t.split(None, 1)[1].replace(' ', '') + ')')
    elif op == 's_barrier': flush(); out.append('BAR')
    elif op.startswith('ds_'): flush(); out.append('D')
    elif op.startswith('buffer_store') or op.startswith('global_store'): flush(); out.append('ST')
    elif op.startswith('v_'): valu += 1
flush()
res = []; prev = None; cnt = 0
for t in out:
    if t == prev: cnt += 1
    else:
        if prev: res.append(prev + ('x%d' % cnt if cnt > 1 else ''))
        prev = t; cnt = 1
res.append(prev + ('x%d' % cnt if cnt > 1 else ''))
print(' '.join(res))
