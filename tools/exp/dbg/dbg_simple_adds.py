"""Where do the two elementwise-add kernels of a data-parallel `simple` step (gradient sinks armed) come from?"""
import sys
import torch
sys.path.insert(0, ".")
from embeddingnet_amd import backbones as B
from embeddingnet_amd.optimizers import KerasOptimizer
from embeddingnet_amd.parallel import GradReducer
from embeddingnet_amd.train_step import TripletTrainer
dev = torch.device("cuda:0")
x = torch.rand((12, 105, 105, 3), device=dev)
flats = []
for direct in (False, True):
    base, _ = B.get_backbone((105, 105, 3), encodings_len=32, backbone_name="simple", backbone_weights=None, seed=2, device=dev)
    params = [p for p in base.parameters() if p.requires_grad]
    names = {p.data_ptr(): n for n, p in base.named_parameters()}
    opt = KerasOptimizer(params, "sgd", 0.0)
    red = GradReducer(params)
    tr = TripletTrainer(base, opt, 4, 3, margin=0.5, negatives_selection_mode="hardest", reducer=red)
    red.direct(direct)
    tr.step(x)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
        tr.step(x)
        torch.cuda.synchronize()
    if direct:
        for e in prof.events():
            if e.name in ("aten::add", "aten::add_"):
                print(e.name, e.input_shapes, [f for f in (e.stack or [])][:6])
    flats.append({names[p.data_ptr()]: red.flat[off:off + n].clone() for p, (off, n) in ((p, red._slot[p]) for p in params)})
    red.close()
for k in flats[0]:
    print(k, "equal" if torch.equal(flats[0][k], flats[1][k]) else (flats[0][k] - flats[1][k]).abs().max().item())
