import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], "ms/step %.4f"%d["ms_per_step"], "records", [round(x,4) for x in r.get("records_kernel_ms",[])], "trace", [round(x,4) for x in r["trace_kernel_ms"]], "shade", [round(x,4) for x in r["shade_kernel_ms"]])
