import json, sys, collections
ref = json.load(open(sys.argv[1]))
for f in sys.argv[2:]:
    d = json.load(open(f))
    bad = [k for k in ref if d.get(k) != ref[k]]
    groups = collections.OrderedDict()
    for k in bad:
        groups.setdefault(".".join(k.split(".")[:3]), []).append(k)
    print(f"{f}: {len(bad)} of {len(ref)} differ: " + "; ".join(f"{g}({len(v)})" for g, v in list(groups.items())[:14]))
    if len(bad) <= 12:
        for k in bad: print("    ", k)
