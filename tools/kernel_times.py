import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import connectome_gnn_amd as C
from connectome_gnn_amd import _lib
from connectome_gnn_amd.resident import assemble_batch
from connectome_gnn_amd.synthetic import generate_packed
ds = generate_packed(512, 360, 14, seed=1).to("cuda")
b = assemble_batch(ds, torch.arange(4096) % 512)
torch.manual_seed(0)
m = C.GCNConnectome(5, 64, impl="fused").to("cuda").train()
names = ["cgnn_gcn_l0_fwd", "cgnn_gcn_l0_bwd", "cgnn_gcn_fused_fwd", "cgnn_gcn_fused_bwd", "cgnn_gcn_fused_pool_fwd"]
for _ in range(3): m(b).sum().backward()
_lib.TIMER = _lib.KernelTimer(names)
for _ in range(10): m(b).sum().backward()
torch.cuda.synchronize()
for n in names:
    v = _lib.TIMER.ms(n); print(n, round(sum(v)/len(v)*1e3, 1), "us")
