import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import torch, synthetic_layers as SL
amd = importlib.import_module("compute-engine_amd")
layer = SL.Layer(256, 56, 56, 256, 3, 3, 256, padding=SL.PADDING_SAME, pad_values=1)
w, mul, bias, thr = SL.weights(layer, 3)
x = torch.from_numpy(SL.activations(layer, 4)).to("cuda:0")
plan = amd.Bconv2dPlan(layer.params(amd, amd.F32, 1.0, 0)); plan.set_weights(w, mul, bias, None)
out = plan.run(x); torch.cuda.synchronize()
t0 = time.perf_counter(); res = []
while time.perf_counter() - t0 < 3.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): plan.run(x, out)
    e1.record(); torch.cuda.synchronize()
    res.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 100))
print(" ".join("%.2fs:%.4f" % r for r in res[::4]))
