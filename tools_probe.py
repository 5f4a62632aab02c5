import sys, json, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda', 0)
for B in (40, 80, 85, 160, 170, 256, 341):
    r = bench.roofline_leg(dev, B=B)
    print(B, 3*B, r['avg_launch_us'], r['frac'])
