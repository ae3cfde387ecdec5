import csv, sys, glob, collections
tag=sys.argv[1]; pat=sys.argv[2] if len(sys.argv)>2 else 'conv'
for d in sorted(glob.glob(f'gpurun_out/{tag}_*/r_counter_collection.csv')):
    rows=list(csv.DictReader(open(d)))
    acc=collections.defaultdict(list)
    for r in rows:
        if pat in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:50], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in acc.items():
        print(f"{d.split('/')[1]:12s} {k:50s} {c:32s} n={len(v):3d} last={v[-1]:.4g}")
