import sys, torch
sys.path.insert(0, '.')
import connectome_gnn_amd as C
from connectome_gnn_amd import ops
torch.manual_seed(0)
graphs = []
n, deg = 1000, 50            # uniform random: 5 % density, ~26 entries per fragment
for i in range(64):
    src = torch.randint(0, n, (n * deg,))
    dst = torch.randint(0, n, (n * deg,))
    keep = src != dst
    ei = torch.stack([src[keep], dst[keep]])
    graphs.append(C.ConnectomeGraph(torch.randn(n, 5), ei, torch.rand(ei.shape[1]), torch.tensor(0)))
b = C.collate_graphs(graphs).to('cuda')
s = b.structure()
norm = s.gcn_norm()
x16 = torch.randn(s.num_nodes, 256, device='cuda').half()
m = ops.dense_adj_f16(s, norm.coef_dst, norm.selfc)
pk = ops.dense_pack_f16(s, norm.coef_dst, norm.selfc)
print('nnz', pk.nnz, 'dense', pk.num_dense, 'sparse', pk.num_sparse, 'MB', pk.nbytes()/1e6)
y0 = ops.dense_aggregate_f16_raw(s, m, x16); y1 = ops.dense_aggregate_c16_raw(s, pk, x16)
print('equal', torch.equal(y0, y1), float((y0.float() - y1.float()).abs().max()))
junk = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
for name, fn in (('dense ', lambda: ops.dense_aggregate_f16_raw(s, m, x16)), ('packed', lambda: ops.dense_aggregate_c16_raw(s, pk, x16))):
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold: junk.fill_(1)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); e.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(e) * 1e3)
        ts = sorted(ts[2:]); print(name, 'cold' if cold else 'warm', round(ts[len(ts) // 2], 1), 'us')
