import csv, glob, collections, sys
rows=[]
for f in glob.glob("gpurun_out/pmc_ablate/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trim_kernel" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), "scan" if "0>(" in r["Kernel_Name"] else "resolve", r["Counter_Name"], float(r["Counter_Value"])))
rows.sort()
# group by dispatch
by=collections.OrderedDict()
for d,k,c,v in rows:
    by.setdefault((d,k),{})[c]=v
seq=list(by.items())
# 7 launches per variant, each launch = scan+resolve dispatch
names=["full","full_nofilter","no_polyA","no_qtrim","no_5prime","no_3prime","no_cuts","only_5prime","only_3prime","only_3prime_mo10","only_5prime_mo3","only_poly","only_cuts","only_qtrim"]
scan=[x for x in seq if x[0][1]=="scan"]; res=[x for x in seq if x[0][1]=="resolve"]
print(len(scan),len(res))
for i,nm in enumerate(names):
    s=scan[i*7+6][1] if i*7+6 < len(scan) else {}
    r=res[i*7+6][1] if i*7+6 < len(res) else {}
    print(f"{nm:18s} scan VALU {s.get('SQ_INSTS_VALU',0)/1e6:8.1f}M SALU {s.get('SQ_INSTS_SALU',0)/1e6:7.1f}M LDS {s.get('SQ_INSTS_LDS',0)/1e6:6.1f}M | resolve VALU {r.get('SQ_INSTS_VALU',0)/1e6:7.1f}M")
