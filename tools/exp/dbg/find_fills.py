"""Where do the torch fill / copy launches of a training step come from?  (torch profiler, CPU op -> Python stack)"""
import sys, collections, torch
sys.path.insert(0, '.')
from embeddingnet_amd import backbones as B
from embeddingnet_amd.optimizers import KerasOptimizer
from embeddingnet_amd.train_step import TripletTrainer
dev = torch.device('cuda', 0)
base, _ = B.get_backbone(input_shape=(224, 224, 3), encodings_len=256, backbone_name="resnet18", embeddings_normalization=True, backbone_weights=None, seed=5)
base.to(dev)
tr = TripletTrainer(base, KerasOptimizer(base.parameters(), "radam", 1e-4), k_classes=8, k_samples=4, margin=0.5, negatives_selection_mode="hardest", graph=False)
imgs = torch.rand(32, 224, 224, 3, device=dev)
for _ in range(3): tr.step(imgs)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(imgs); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::ones", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::sum", "aten::div"):
        st = []; q = e.cpu_parent
        while q is not None and len(st) < 4:
            st.append(q.name); q = q.cpu_parent
        cnt[(e.name + str(list(e.input_shapes) if e.input_shapes else ""), tuple(st))] += 1
for (n, st), c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(c, n, " <- ".join(s.split("/")[-1] for s in st))
