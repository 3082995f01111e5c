import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import make_golden_superpoint as G
from lightglue_amd import SuperPoint
prec = sys.argv[1]
model = SuperPoint(weights=G.encoder_state_dict(0), max_num_keypoints=2048, conv_precision=prec).cuda().eval()
img = torch.rand(8, 1, 480, 640, device="cuda")
for _ in range(3): model.encode(img)
torch.cuda.synchronize()
