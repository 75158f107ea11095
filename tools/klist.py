import sys,json
d=json.loads(sys.stdin.read())
print(d["value"], d["ms_per_step"])
for t,v in d["kernels"].items():
    w=v["work_per_launch"]; us=v["avg_us"]
    print("%-60s %8.1f us x %5.1f  %8.3f ms/step  %s" % (t, us, v["launches_per_step"], v["ms_per_step"], ("%.1f TF" % (w/us/1e6)) if "gemm" in t else ""))
