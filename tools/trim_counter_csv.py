"""Keep the rows of a rocprofv3 counter_collection.csv that belong to our kernels (the file also lists every memset)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "drt::" in r["Kernel_Name"]]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else [])
    w.writeheader()
    for r in keep:
        r["Kernel_Name"] = r["Kernel_Name"].split("(drt::SceneView")[0]
        w.writerow(r)
