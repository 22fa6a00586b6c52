import sys, re, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for l in sys.stdin:
    m = re.match(r"(forward|backward_data|backward_weight)\s", l)
    if not m: continue
    parts = l.split("|")[1].split()
    agg[m.group(1)][0] += int(parts[0]); agg[m.group(1)][1] += float(parts[1])
for k, (n, ms) in agg.items(): print(f"{k:16s} launches {n:4d}  ms {ms:7.2f}")
