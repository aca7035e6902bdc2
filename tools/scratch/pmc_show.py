import csv, collections, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/dcn_pmc_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'dcn' not in k: continue
        k = k.split('(anonymous namespace)::')[-1].split('(')[0][:28]
        if 'PlanEntry' in k: k = 'dgrad_gather/plan'
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
want = sys.argv[1:] 
for k, d in acc.items():
    if want and not any(w in k for w in want): continue
    print(k)
    for c, v in sorted(d.items()):
        print('   %-32s %16.0f' % (c, sum(v) / len(v)))
