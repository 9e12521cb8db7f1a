import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if not any(s in k for s in ("encoder", "predictor", "gemm", "head_out")): continue
    print(k)
    for c, v in d.items():
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
