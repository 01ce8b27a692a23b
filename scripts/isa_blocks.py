"""Basic-block instruction counts of one kernel in a hipcc -save-temps .s file (loop bodies)."""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
L = open(path).read().split("\n")
start = next(i for i, l in enumerate(L) if re.match(r"^_Z\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
cur, cnt, sizes, ops = "entry", 0, [], collections.Counter()
for l in L[start:end]:
    t = l.strip()
    if re.match(r"\.LBB\d+_\d+:", t):
        sizes.append((cur, cnt, ops))
        cur, cnt, ops = t.split(":")[0] + " " + t.split(";")[-1][:60], 0, collections.Counter()
    elif t and not t.startswith((";", ".", "//")):
        cnt += 1
        ops[t.split()[0]] += 1
sizes.append((cur, cnt, ops))
print("total", sum(c for _, c, _ in sizes))
for n, c, o in sizes:
    if c > int(sys.argv[3]) if len(sys.argv) > 3 else 80:
        print("%5d  %s" % (c, n))
        print("       ", o.most_common(8))
