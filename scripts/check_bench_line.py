import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
need=["metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data","config","roofline","cpu_baseline"]
print("missing:",[k for k in need if k not in j])
r=j["roofline"]; c=j["cpu_baseline"]
print(j["metric"],"|","value %.4g"%j["value"],j["unit"],"ms_per_step %.3f"%j["ms_per_step"],"steps",j["steps"],"warmup",j["warmup"],"dtype",j["dtype"],"vs_baseline",j["vs_baseline"])
print("roofline",{k:r[k] for k in ("bound","achieved","peak","unit","frac","traffic")})
print("cpu",{k:c[k] for k in ("value","unit","cores","kind")},"| sample:",c["sample"][:80])
print("workload:",j["config"]["workload"])
print("frontend:",{k:round(v["run_ms"],2) for k,v in j["frontend"].items() if isinstance(v,dict)})
