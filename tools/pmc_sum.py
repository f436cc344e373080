"""Per-kernel means of rocprofv3 --pmc counters: python3 tools/pmc_sum.py <output dir> <kernel name substring>."""
import csv,sys,glob,collections
d=sys.argv[1]; pat=sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if pat not in k: continue
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVES': n[k]+=1
for k,v in acc.items():
    print(k, 'dispatches',n[k])
    for c,x in sorted(v.items()): print('   ',c, x/max(1,n[k]))
