import json,sys
for f in sys.argv[1:]:
    d=json.loads(open("gpurun_out/"+f+".log").read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["roofline"]["achieved"],1), round(d["mask_iou"],5), {k:round(v["ms_per_step"],3) for k,v in d["stages"].items()})
