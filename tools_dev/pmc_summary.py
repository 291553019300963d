# developer tool: average rocprofv3 --pmc counters per kernel from a counter_collection csv
import sys, csv, re, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        m=re.search(r'(\w+_kernel(<[^>]*>)?)',row['Kernel_Name']); k=m.group(1) if m else row['Kernel_Name'][:40]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k,c in acc.items():
    print(k, {n:(sum(v)/len(v)) for n,v in c.items()}, 'launches', max(len(v) for v in c.values()))
