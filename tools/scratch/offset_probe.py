import sys, torch
sys.path.insert(0, '.')
import connectome_gnn_amd as C
sys.path.insert(0, 'tests')
import parity as P
for offset, scale in ((10.0, 1.0), (30.0, 1.0), (100.0, 1.0)):
    b = C.collate_graphs(C.generate_dataset(12, 360, 14, seed=5))
    b.node_features = b.node_features * scale + offset
    torch.manual_seed(11)
    m = C.GCNConnectome(5, 64, dropout=0.0)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    lo, loss_o, g32, st = P.oracle_run("gcn", sd0, b)
    m = m.to("cuda").train()
    lg = m(b.to("cuda"))
    sd = m.state_dict()
    rv = (sd["batch_norms.0.running_var"].cpu() - st["batch_norms.0.running_var"]).abs() / st["batch_norms.0.running_var"].abs()
    print(offset, "logit max abs diff", float((lg.cpu() - lo).abs().max()), "of", float(lo.abs().max()),
          " bn0 running_var max rel diff", float(rv.max()))
