"""Static instruction count of a kernel per source line (line tables): python isa_by_line.py file.s kernel_prefix source.hip [min]
Build the .s with: hipcc -O3 ... -S --cuda-device-only -gline-tables-only."""
import collections
import re
import sys

asm, kernel, source = sys.argv[1:4]
floor = int(sys.argv[4]) if len(sys.argv) > 4 else 8
lines = open(asm).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(kernel)][0]
end = [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][0] + start
files, cnt, cur = {}, collections.Counter(), (0, 0)
for i, l in enumerate(lines[:end]):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = m.group(3) or m.group(2)
        continue
    if i < start:
        continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    s = l.strip()
    if not s or s.startswith(('.', ';', '//')) or s.endswith(':'):
        continue
    cnt[cur] += 1
byfile = collections.Counter()
for (f, ln), c in cnt.items():
    byfile[files.get(f, str(f))] += c
print('instructions per file:', byfile.most_common(6))
base = source.split('/')[-1]
mine = [f for f, n in files.items() if n.endswith(base)]
src = open(source).read().split('\n')
tot = collections.Counter()
for (f, ln), c in cnt.items():
    if f in mine:
        tot[ln] += c
for ln in sorted(tot):
    if tot[ln] >= floor:
        print(f'{ln:5d} {tot[ln]:5d}  {src[ln - 1].strip()[:120]}')
