import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.pop("roofline"); others=r.pop("others",[])
print(d["value"], "patches/s", d["ms_per_step"], "ms/step")
rows=[r]+others
tot=0
for o in rows:
    tot+=o["total_ms"]
    print(f'{o["category"]:18s} {o["bound"]:5s} frac {o["frac"]:.3f} avg_us {o["avg_launch_us"]:8.1f} ms/step {o["total_ms"]/d["steps"]:7.3f} hbm {o["hbm_GBs"]:7.0f} GB/s mfma {o["mfma_TFs"]:7.1f} TF/s')
print("sum ms/step", tot/d["steps"])
