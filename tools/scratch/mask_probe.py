import sys, torch
sys.path.insert(0, ".")
import connectome_gnn_amd as C
from connectome_gnn_amd import ops
b = C.collate_graphs(C.generate_dataset(24, 84, 8, seed=321)).to("cuda")
for impl in ("auto", "layered"):
    torch.manual_seed(11)
    m = C.GCNConnectome(5, 64, dropout=0.3, impl=impl).to("cuda").train()
    m.record_dropout = True
    lg = m(b)
    rec = m.last_dropout
    print(impl, m.impl_used, list(rec.keys()), [x.shape for x in rec["layers"]])
    for x in rec["layers"]:
        bits = torch.stack([(x >> i) & 1 for i in range(4)]).float().mean()
        print("  before bwd: raw bit mean", float(bits), "unpack mean", float(ops.unpack_keep_bits(x, b.num_nodes, 64).mean()),
              "byte hist", torch.bincount(x.long(), minlength=16).tolist())
    lg.sum().backward()
    for x in rec["layers"]:
        print("  after bwd: unpack mean", float(ops.unpack_keep_bits(x, b.num_nodes, 64).mean()))
    print("  head", float((rec["head_factor"] > 0).float().mean()))
