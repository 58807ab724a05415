import json
for f in ("forced_1","single_1","forced_2","single_2"):
    d=json.loads(open("gpurun_out/r05j_%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["pcg"]["mean_iterations"], d["host"], d["pcg"].get("batch_prediction"))
