import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, gan2shape_amd
from gan2shape_amd.model import GAN2Shape
from gan2shape_amd.trainer import Trainer
from gan2shape_amd.graphs import GraphedSteps
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(dev))
torch.manual_seed(0)
tr = Trainer(GAN2Shape, bench.face_config(8), device=dev, capturable=True)
image, latent = bench.synthetic_sample(tr.model, 1234, dev)
r = bench.StepRunner(tr, image, latent)
for k in (1, 2, 3):
    r.run(k)
g = GraphedSteps(tr, image, latent); g.collected = dict(r.collected)
for k in (1, 2, 3):
    g.capture(k)
gr = bench.GraphedRunner(g)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    for k in (1, 2, 3):
        print(f"rep {rep} step {k}: eager {t(lambda: r.run(k)):.2f} ms   graph {t(lambda: gr.run(k)):.2f} ms", flush=True)
print("--- transitions (graph mode)")
def seq(pattern, reps):
    for k in pattern: gr.run(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for k in pattern: gr.run(k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (reps * len(pattern)) * 1e3
grouped = [1]*7 + [2]*7 + [3]*6
inter = [1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2]
print("grouped 7:7:6 ms/step", seq(grouped, 4))
print("interleaved   ms/step", seq(inter, 4))
print("1,2 alternating", seq([1, 2], 20), " expected", (3.39 + 13.52) / 2)
# host-side launch cost
t0 = time.perf_counter()
for _ in range(10): gr.run(3)
t_host = (time.perf_counter() - t0) / 10 * 1e3
torch.cuda.synchronize()
print("host time per graph launch (step 3), ms:", t_host)
