"""How fast are the library's strided-batched GEMMs at the shapes of an aggregate-then-project GAT input layer?
agg [n, H, F] (per head: the attention-weighted sum of the RAW feature rows), W [H, D, F]:
  fwd   out[h] = agg[:, h, :] @ W[h].T          [n, F] x [F, D]
  dagg  dagg[h] = g[:, h, :] @ W[h]             [n, D] x [D, F]
  dW    dW[h] = g[:, h, :].T @ agg[:, h, :]     [D, n] x [n, F]   (K = n: needs a split over n)
Run on the GPU box: python profiles/gat_bmm_probe.py"""
import torch, time
n, H, F, D = 51200, 8, 100, 32
dev = "cuda"
agg = torch.randn(n, H, F, device=dev)
W = torch.randn(H, D, F, device=dev)
g = torch.randn(n, H, D, device=dev)
def t(f, it=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
A = agg.permute(1, 0, 2)      # [H, n, F] strided
G = g.permute(1, 0, 2)        # [H, n, D] strided
print("fwd  bmm strided   %.1f us" % t(lambda: torch.bmm(A, W.transpose(1, 2))))
print("dagg bmm strided   %.1f us" % t(lambda: torch.bmm(G, W)))
print("dW   bmm strided   %.1f us" % t(lambda: torch.bmm(G.transpose(1, 2), A)))
Ac, Gc = A.contiguous(), G.contiguous()
print("fwd  bmm contig    %.1f us" % t(lambda: torch.bmm(Ac, W.transpose(1, 2))))
print("dagg bmm contig    %.1f us" % t(lambda: torch.bmm(Gc, W)))
print("dW   bmm contig    %.1f us" % t(lambda: torch.bmm(Gc.transpose(1, 2), Ac)))
print("einsum fwd         %.1f us" % t(lambda: torch.einsum("nhf,hdf->nhd", agg, W)))
# dW with an explicit split over n (S slabs -> [S*H, D, F], then a sum)
for S in (16, 64):
    A2 = agg.view(S, n // S, H, F).permute(0, 2, 1, 3).reshape(S * H, n // S, F) if False else None
    def dw_split():
        a = agg.view(S, n // S, H, F).permute(0, 2, 1, 3)      # [S, H, n/S, F] strided
        gg = g.view(S, n // S, H, D).permute(0, 2, 3, 1)       # [S, H, D, n/S]
        return torch.matmul(gg, a).sum(0)
    print("dW split %3d       %.1f us" % (S, t(dw_split)))
# the dense alternative for scale: project all 500k source rows
x = torch.randn(500_000, F, device=dev); Wd = W.reshape(H * D, F)
print("dense projection of 500k rows  %.1f us" % t(lambda: x @ Wd.t(), 20))
