"""Instruction mix of every loop of one kernel in a --save-temps .s file: isa_loops.py file.s mangled_name"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read()
name = sys.argv[2]
i = txt.index("\n" + name + ":"); j = txt.index("s_endpgm", i)
lines = txt[i:j].split("\n")
labels = {}
for n, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = n
for n, l in enumerate(lines):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        a = labels[m.group(1)]
        body = [x.strip() for x in lines[a:n + 1] if x.strip() and not x.strip().startswith(('.', ';'))]
        c = Counter()
        for x in body:
            op = x.split()[0]
            k = ('fma64' if op.startswith(('v_fma_f64', 'v_fmac_f64')) else 'addmul64' if op.startswith(('v_add_f64', 'v_mul_f64'))
                 else 'cvt' if op.startswith('v_cvt') else 'vmov' if op.startswith('v_mov') else 'v_other' if op.startswith('v_')
                 else 'waitcnt' if op.startswith('s_waitcnt') else 'nop' if op.startswith('s_nop') else 'salu' if op.startswith('s_') else op)
            c[k] += 1
        print(a, n, len(body), dict(c))
