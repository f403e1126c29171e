import torch, time, numpy as np, sys
sys.path.insert(0, '.')
from coupe.optical_flow_based_deep_video_stabilization_amd import postfilters as pf
for (B,H,W) in [(8,512,512),(1,1080,1920),(1,720,1280)]:
    f = torch.randn(B,H,W,2,device='cuda')*3
    img = torch.randint(0,256,(B,H,W,3),dtype=torch.uint8,device='cuda')
    for K in (128,256):
        for _ in range(3): Hm,_ = pf.find_homography(f,K=K)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(20): Hm,_ = pf.find_homography(f,K=K)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
        print(B,H,W,'K',K,'fit ms',round(dt*1e3,3))
    eye = torch.eye(3,dtype=torch.float64,device='cuda').expand(B,3,3).contiguous()
    for _ in range(3): o = pf.warp_perspective_u8(img, eye)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): o = pf.warp_perspective_u8(img, eye)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    print(B,H,W,'warp ms',round(dt*1e3,3),'GB/s',round(B*H*W*6/dt/1e9,1))
