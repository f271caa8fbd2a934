#!/usr/bin/env python3
"""Static loop map of one kernel in a -save-temps .s file: every backward branch with the instruction mix of the
span it closes (nested loops show up as nested spans).  usage: isa_loops.py file.s mangled_prefix"""
import re
import sys

def main():
    s = open(sys.argv[1]).read().split('\n')
    start = [i for i, l in enumerate(s) if l.startswith(sys.argv[2])][0]
    end = [i for i, l in enumerate(s) if i > start and l.strip().startswith('s_endpgm')][0]
    labels, ins = {}, []
    for l in s[start:end + 1]:
        t = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        ins.append(t)
    print('total instructions', len(ins))
    for i, t in enumerate(ins):
        m = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', t)
        if m and m.group(1) in labels and labels[m.group(1)] <= i:
            a = labels[m.group(1)]
            seg = ins[a:i + 1]
            cnt = lambda p: sum(1 for x in seg if x.startswith(p))
            print('loop %-12s [%5d..%5d] len %5d  valu %5d  ds %4d  salu %4d  vmem %3d' %
                  (m.group(1), a, i, i - a + 1, cnt('v_'), cnt('ds_'), cnt('s_'), cnt('buffer_') + cnt('global_')))

main()
