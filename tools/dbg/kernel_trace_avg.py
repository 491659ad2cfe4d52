import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:44],r.get("Grid_Size_X") or r.get("Grid_Size"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print(k,len(v),"avg %.1f us min %.1f"%(sum(v)/len(v),min(v)))
