import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# long kernels
long=[r for r in rows if int(r["End_Timestamp"])-int(r["Start_Timestamp"])>5e6]
print("kernels > 5 ms:",len(long))
for r in long[:10]:
    print("  ",(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,"ms",r["Kernel_Name"][:80],"queue",r.get("Queue_Id"))
# gaps > 10 ms between consecutive kernels
prev=None;gaps=[]
for r in rows:
    if prev is not None:
        g=int(r["Start_Timestamp"])-int(prev["End_Timestamp"])
        if g>10e6: gaps.append((g/1e6,prev["Kernel_Name"][:50],r["Kernel_Name"][:50],r.get("Queue_Id")))
    prev=r
print("gaps > 10 ms:",len(gaps))
for g in gaps[:12]: print("  ",g)
