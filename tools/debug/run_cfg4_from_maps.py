import sys, json, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))); sys.path.insert(0, sys.path[0] + '/tools')
import bench_workloads as bw
print(json.dumps(bw.run_cfg4_from_maps(torch.device('cuda')), indent=1))
