"""Stress check of the BatchNorm tail (csrc/bn_tail.h): over many training steps the loss, every parameter gradient
and the running statistics must be (a) bit-identical between two runs of the tail (the fixed-point sums are
order-independent; the last-workgroup hand-over has no fence: a race would show as a run-to-run difference) and
(b) within rounding of the slab protocol's (whose fp64 tree fold of 256 rounded partials differs from the exact
fixed-point sum in the last bits: a coefficient is 1 fp32 ulp off now and then, which a training run amplifies slowly).
   python tools/tail_stress.py [steps] [graphs] [rois]"""
import sys

import torch

sys.path.insert(0, ".")
import connectome_gnn_amd as C                          # noqa: E402
from connectome_gnn_amd import fused                     # noqa: E402
from connectome_gnn_amd.resident import assemble_batch   # noqa: E402
from connectome_gnn_amd.synthetic import generate_packed  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
graphs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rois = int(sys.argv[3]) if len(sys.argv) > 3 else 84
k = 8 if rois < 200 else 14
ds = generate_packed(graphs, rois, k, seed=3).to("cuda")
g = torch.Generator().manual_seed(0)
batches = [assemble_batch(ds, torch.randperm(graphs, generator=g)) for _ in range(4)]
for b in batches:
    b.structure()


def run(no_tails):
    fused._NO_TAILS = no_tails
    torch.manual_seed(1)
    m = C.GCNConnectome(5, 64, dropout=0.3).to("cuda").train()
    opt = torch.optim.SGD(m.parameters(), lr=1e-2)
    out = []
    for i in range(steps):
        torch.manual_seed(100 + i)                      # same dropout masks both ways
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(m(batches[i % 4]), batches[i % 4].labels)
        loss.backward()
        opt.step()
        # (a GCN bias ahead of BatchNorm has a zero true gradient: its .grad is rounding noise, compared run to
        # run only)
        out.append((loss.detach().clone(), [p.grad.clone() for p in m.parameters()],
                    [bn.running_var.clone() for bn in m.batch_norms],
                    [p.grad.clone() for n_, p in m.named_parameters() if not (n_.startswith("convs.") and n_.endswith(".bias"))]))
    torch.cuda.synchronize()
    return out


a, b, c = run(False), run(False), run(True)
bad = 0
for i, (x, y) in enumerate(zip(a, b)):
    same = torch.equal(x[0], y[0]) and all(torch.equal(p, q) for p, q in zip(x[1], y[1])) \
        and all(torch.equal(p, q) for p, q in zip(x[2], y[2]))
    if not same:
        bad += 1
        if bad <= 3:
            print("step", i, "differs run to run: loss", float(x[0]), float(y[0]))
worst = 0.0
for x, y in zip(a[:3], c[:3]):                            # (later steps: the trajectories drift apart slowly)
    for p, q in zip(x[3] + x[2], y[3] + y[2]):
        worst = max(worst, float((p - q).abs().max()) / max(float(q.abs().max()), 1e-30))
print(f"{steps} steps of {graphs} x {rois}: {bad} steps differ between two runs of the tail; "
      f"tail vs slab protocol over the first 3 steps: worst relative difference {worst:.2e}")
sys.exit(1 if (bad or worst > 1e-4) else 0)
