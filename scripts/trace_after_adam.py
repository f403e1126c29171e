import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
i0 = ad[-3]
t0 = int(rows[i0]["Start_Timestamp"])
n = 0
for r in rows[i0:i0 + 75]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:60]}")
