"""SiameseTrainer (embeddingnet_amd/train_step.py): the Siamese step of the reference (/root/reference/embedding_net/models.py:192-236,
tools/train.py:108-119) behind TripletTrainer's machinery — step context, deferred slab sums, one-launch optimizer, HIP graph.

  * its eager steps are bit for bit the plain loop's (zero_grad / model / contrastive_loss / backward / optimizer.step);
  * a captured step replays to the same losses and weights, bit for bit, 'l2' and 'l1' heads;
  * graph='auto' leaves a host-work probe; nothing is left in the step context;
  * (C3's parity against the float64 oracle is tests/test_full_size_gpu.py::test_c3_siamese_resnet50_step_vs_oracle — that
    test drives the same model through the plain loop, which the first bullet ties to the trainer).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _net(dev, distance_type="l2", backbone="resnet18", image=64, enc=32, seed=4):
    from embeddingnet_amd.models import SiameseNet
    torch.manual_seed(11)            # (the heads' Dense layers draw from torch's global generator)
    return SiameseNet({"model": dict(input_shape=[image, image, 3], encodings_len=enc, mode="siamese", distance_type=distance_type,
                                     backbone_name=backbone, backbone_weights=None, freeze_backbone=False,
                                     embeddings_normalization=True, device=dev, seed=seed),
                       "dataloader": {}, "generator": {}, "train": {},
                       "general": {"work_dir": "work_dirs/", "project_name": "t"}}, training=True)


def _data(dev, pairs, image, steps, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    xs = [(torch.rand((pairs, image, image, 3), device=dev, generator=gen), torch.rand((pairs, image, image, 3), device=dev, generator=gen))
          for _ in range(steps)]
    y = (torch.arange(pairs, device=dev) < pairs // 2).float().reshape(-1, 1)
    return xs, y


def _params(net):
    return [p for p in net.model.parameters() if p.requires_grad]


def _run_trainer(dev, distance_type, graph, steps=14, rule="adam"):
    from embeddingnet_amd.losses_and_accuracies import contrastive_loss
    from embeddingnet_amd.optimizers import KerasOptimizer
    from embeddingnet_amd.train_step import SiameseTrainer
    net = _net(dev, distance_type)
    net.model.train()
    tr = SiameseTrainer(net.model, KerasOptimizer(_params(net), rule, 1e-3), contrastive_loss, graph=graph)
    xs, y = _data(dev, 8, 64, steps, seed=3)
    losses = [float(tr.step(x1, x2, y)) for x1, x2 in xs]
    assert not tr.ctx.leftovers(), tr.ctx.leftovers()
    return losses, [p.detach().clone() for p in _params(net)], tr


@pytest.mark.parametrize("distance_type", ["l2", "l1"])
def test_trainer_steps_equal_the_plain_loop(dev, distance_type):
    from embeddingnet_amd.losses_and_accuracies import contrastive_loss
    from embeddingnet_amd.optimizers import KerasOptimizer
    steps = 4
    got, w_got, _ = _run_trainer(dev, distance_type, graph=False, steps=steps)
    net = _net(dev, distance_type)
    net.model.train()
    opt = KerasOptimizer(_params(net), "adam", 1e-3)
    xs, y = _data(dev, 8, 64, steps, seed=3)
    want = []
    for x1, x2 in xs:
        opt.zero_grad(set_to_none=True)
        loss = contrastive_loss(y, net.model([x1, x2])[0])
        loss.backward()
        opt.step()
        want.append(float(loss))
    assert got == want, (got, want)
    for a, b in zip(w_got, _params(net)):
        assert torch.equal(a, b.detach())


@pytest.mark.parametrize("distance_type", ["l2", "l1"])
def test_captured_step_equals_eager_steps(dev, distance_type):
    eager, w_eager, _ = _run_trainer(dev, distance_type, graph=False)
    replay, w_replay, tr = _run_trainer(dev, distance_type, graph=True)
    assert tr._graph is not None, getattr(tr, "_graph_error", None)            # the step after GRAPH_WARMUP was captured
    assert eager == replay, (eager, replay)
    for a, b in zip(w_eager, w_replay):
        assert torch.equal(a, b)


def test_auto_mode_probes_the_host_work(dev):
    losses, _, tr = _run_trainer(dev, "l2", graph="auto", steps=12)
    assert all(l == l and l > 0 for l in losses)
    pr = tr.graph_probe
    assert pr["eager_ms"] > 0 and pr["host_ms"] > 0
    # a 64x64 ResNet18 pair step is launch-bound: the probe keeps the graph (or records the replay that lost)
    assert tr._graph is not None or "replay_ms" in pr or pr["host_ms"] < 0.8 * pr["eager_ms"], pr
