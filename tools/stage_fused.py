import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import gpu_util
from lightglue_amd import synthetic as synth
sd = synth.make_state_dict(0, recipe="A"); data = synth.make_batch(7, 2, 200, 160)
for prec in sys.argv[1:]:
    res = gpu_util.stage_errors(sd, data, prec, dict(depth_confidence=-1, width_confidence=-1), fused=True)
    print(prec, {k: '%.2e'%v[1] for k,v in res.items()})
