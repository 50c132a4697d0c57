#!/usr/bin/env python3
"""isa_loops.py <kernel.s>: list the loops (backward branches) of one kernel's gfx950 assembly with their instruction mix
(VALU / SALU / LDS / VMEM / other), so that a change to a hot loop can be priced before it goes to the GPU."""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
labels = {}
for i, ln in enumerate(lines):
    m = re.match(r'^(\.LBB[0-9_]+):', ln)
    if m:
        labels[m.group(1)] = i
loops = []
for i, ln in enumerate(lines):
    m = re.match(r'\s+s_(c?branch\w*)\s+(\.LBB[0-9_]+)', ln)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i))


def mix(a, b):
    c = dict(valu=0, salu=0, lds=0, vmem=0, smem=0, other=0, barrier=0, trans=0)
    for ln in lines[a:b + 1]:
        t = ln.strip().split()
        if not t or t[0].endswith(':') or t[0].startswith(('.', ';')):
            continue
        op = t[0]
        if op == 's_barrier':
            c['barrier'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
            if re.match(r'v_(rcp|sqrt|rsq|exp|log|sin|cos|div_)', op):
                c['trans'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
            c['vmem'] += 1
        elif op.startswith(('s_load', 's_buffer_load')):
            c['smem'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
        else:
            c['other'] += 1
    return c


for a, b in sorted(loops):
    c = mix(a, b)
    if c['barrier'] or (b - a) > 40:
        print(f'lines {a + 1:5d}-{b + 1:5d} ({b - a + 1:4d} lines): ' + ' '.join(f'{k}={v}' for k, v in c.items() if v))
