"""Probe for ADVICE r3 (graphed.py): capture a training step while the autograd graph of an EARLIER eager step
on the same parameters is still alive (its `loss` tensor kept).  Run as a child process: undetected, the condition
takes the process down in capture_end (that is how the cause was found: torch's engine prints "The AccumulateGrad
node's stream does not match ..." right before the segmentation fault).
   python tools/capture_probe.py keep   -> GraphedTrainStep must raise RuntimeError, then succeed once the graphs are dropped
   python tools/capture_probe.py drop   -> no graph kept: capture + replay"""
import faulthandler
import sys

import torch

faulthandler.enable()
sys.path.insert(0, ".")
import connectome_gnn_amd as C                                      # noqa: E402
from connectome_gnn_amd import graphed, optim                        # noqa: E402

keep = len(sys.argv) < 2 or sys.argv[1] == "keep"
dev = "cuda"
b = C.collate_graphs(C.generate_dataset(32, 84, 8, seed=1)).to(dev)
torch.manual_seed(0)
m = C.GCNConnectome(5, 64).to(dev).train()
opt = optim.Adam(m.parameters(), lr=1e-3)
crit = torch.nn.CrossEntropyLoss()
held = []
for _ in range(2):
    opt.zero_grad()
    loss = crit(m(b), b.labels)
    loss.backward()
    opt.step()
    if keep:
        held.append(loss)                 # NOT detached: the step's autograd graph (and its AccumulateGrad nodes) lives on
torch.cuda.synchronize()
print("eager steps done; held graphs:", len(held), flush=True)
if keep:
    try:
        graphed.GraphedTrainStep(m, opt, b, warmup=1)
        print("NOT DETECTED", flush=True)
        sys.exit(3)
    except RuntimeError as e:
        assert "autograd graph of an earlier step" in str(e), e
        print("detected:", str(e)[:90], flush=True)
    before = [p.detach().clone() for p in m.parameters()]
    held.clear()
    del loss                                  # the graphs are gone: the same call must now succeed
else:
    del loss
step = graphed.GraphedTrainStep(m, opt, b, warmup=1)
for _ in range(3):
    out = step()
torch.cuda.synchronize()
print("captured + replayed ok, loss", float(out), flush=True)
