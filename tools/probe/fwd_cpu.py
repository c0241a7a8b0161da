import sys, time, torch, cProfile, pstats
sys.path.insert(0, ".")
import bench
from yolopoint_amd.utils.synthetic import synth_image
dev = torch.device("cuda:0")
m, _ = bench.build_model("l", "f16", dev)
m.model.use_graph = True
x = synth_image(1, 3, 1280, 1280, 100).to(dev)
for _ in range(5): m(x)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): m(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
