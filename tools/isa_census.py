"""dev tool: instruction census of the largest backward-branch loop of one kernel in a hipcc -S listing.
usage: python tools/isa_census.py file.s KERNEL_NAME_SUBSTRING"""
import collections, re, sys
s = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*' + re.escape(sys.argv[2]) + r'\S*:', l))
end = next(i for i in range(start, len(s)) if s[i].strip().startswith('s_endpgm'))
lines = [l.strip() for l in s[start:end]]
labels = {re.match(r'^(\.LBB\d+_\d+):', l).group(1): i for i, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
back = []
for i, l in enumerate(lines):
    mm = re.match(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        back.append((labels[mm.group(1)], i))
print("backward branches (label line, branch line):", back, "kernel lines:", len(lines))
a, b = max(back, key=lambda x: x[1] - x[0])
c = collections.Counter()
for l in lines[a:b]:
    if not l or l.startswith(('.', ';')):
        continue
    c[l.split()[0]] += 1
print("loop instructions:", sum(c.values()))
for k, v in c.most_common(60):
    print(f"{v:5d} {k}")
