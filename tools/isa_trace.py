"""compressed instruction-class trace of a kernel's ISA (from `hipcc -S`): tools/isa_trace.py file.s kernel_substring"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
i0 = next(i for i, l in enumerate(txt) if l.startswith('_Z') and key in l and l.split(';')[0].strip().endswith(':'))
i1 = next(i for i in range(i0, len(txt)) if '-- End function' in txt[i])
lines = txt[i0:i1]
def cls(t):
    op = t.split()[0]
    if op.startswith('v_mfma'): return 'MFMA'
    if op.startswith('v_exp'): return 'exp'
    if op.startswith('ds_read'): return 'dsr'
    if op.startswith('ds_'): return op
    if op.startswith('global_load_lds'): return 'DMA'
    if op.startswith(('global_', 'buffer_', 'scratch_')): return op
    if op.startswith('v_'): return 'v'
    if op.startswith('s_waitcnt'): return '{' + t.replace('s_waitcnt ', '') + '}'
    if op.startswith('s_barrier'): return 'BARRIER'
    if op.startswith('s_cbranch'): return 'br'
    if op.startswith('s_'): return 's'
    return op
out, prev, n = [], None, 0
for l in lines:
    t = l.split(';')[0].strip()
    if not t or t.startswith('.') and not t.endswith(':'): continue
    c = '\n[' + t + ']' if t.endswith(':') else cls(t)
    if 'Loop Header' in l: c = '\n=== LOOP ' + c
    if c == prev: n += 1
    else:
        if prev: out.append(f"{prev}x{n}" if n > 1 else prev)
        prev, n = c, 1
out.append(prev)
print(' '.join(out))
