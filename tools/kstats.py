"""Dev tool: print a rocprofv3 kernel_stats.csv compactly (kernel names contain commas)."""
import csv, re, sys
for path in sys.argv[1:]:
    print("==", path)
    for r in csv.DictReader(open(path)):
        n = r["Name"]
        if not ("amdrec" in n or "rocclr" in n or "at::native" in n):
            continue
        epi = re.search(r"amdrec::(Epi\w+)", n)
        shape = re.search(r"Shape<([^>]*)>", n)
        short = re.sub(r"[(<].*", "", n.replace("void ", ""))[:48]
        print(f"{short:48s} {epi.group(1) if epi else '':16s} {shape.group(1) if shape else '':22s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f}")
