import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"]: print(sys.argv[2], "calls", r["Calls"], "total ms", round(int(r["TotalDurationNs"])/1e6,3), "avg us", round(float(r["AverageNs"])/1e3,1))
