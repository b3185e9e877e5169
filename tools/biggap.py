import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("tsd::", "")[:40], r.get("Queue_Id"), r.get("Stream_Id", "?")))
mc = glob.glob(os.path.join(sys.argv[1], "**", "*memory_copy_trace.csv"), recursive=True)
for m in mc:
    for r in csv.DictReader(open(m)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy_" + r.get("Direction", "?"), "-", "-"))
rows.sort()
# steady part only
rows = rows[len(rows) // 3:]
gaps = sorted(((rows[i + 1][0] - max(x[1] for x in rows[max(0, i - 6):i + 1]), i) for i in range(len(rows) - 1)), reverse=True)[:3]
for g, i in gaps:
    print("gap %.1f us after row %d" % (g / 1e3, i))
    base = rows[i][1]
    for r in rows[max(0, i - 12):i + 10]:
        print("   %10.1f .. %10.1f  q%s s%s %s" % ((r[0] - base) / 1e3, (r[1] - base) / 1e3, r[3], r[4], r[2]))
