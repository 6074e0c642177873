"""Pretty-print an exp11 output: best-of-rounds per case, base vs new."""
import collections
import re
import sys
d = collections.OrderedDict()
cur = None
for l in open(sys.argv[1]):
    if l.startswith('##'):
        cur = l.split()[1]
        continue
    m = re.match(r'(.*?)\s+([\d.]+) ms\s+([\d.]+) TF/s', l)
    if m:
        d.setdefault(m.group(1).strip(), {}).setdefault(cur, []).append(float(m.group(2)))
for k, v in d.items():
    b, n = min(v['base']), min(v['new'])
    print(f"{k:70s} base {b:7.3f}  new {n:7.3f}  {100 * (b / n - 1):+5.1f}%")
