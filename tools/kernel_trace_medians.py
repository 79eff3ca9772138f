import csv,sys,glob,collections,statistics
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:12]:
    print(f"{k:70s} n={len(v):5d} sum={sum(v)/1e3:8.2f} ms  med={statistics.median(v):8.1f} us  p90={sorted(v)[int(0.9*len(v))]:8.1f}")
